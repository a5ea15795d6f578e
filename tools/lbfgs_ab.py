"""A/B of builds of libgem_hip.so on the same workload (runs ON THE GPU BOX): per-window results and statistics side by side.
   python tools/lbfgs_ab.py libA.so libB.so [...] -- [bench args]"""
import json, os, subprocess, sys
import numpy as np

argv = sys.argv[1:]
libs, extra = (argv[:argv.index("--")], argv[argv.index("--") + 1:]) if "--" in argv else (argv, [])
extra = extra or ["--workload", "configs3", "--windows", "2048", "--vae", "structured"]
out = []
for n, lib in enumerate(libs):
    dump = "/tmp/ab_%d.npz" % n
    env = dict(os.environ, GEM_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--cpu-windows", "0", "--dump", dump] + extra,
                       env=env, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    o = dict(np.load(dump))
    f = o["stats_finished"].astype(bool)
    bad = np.flatnonzero(~f)
    print("%d %-24s all_finished=%s value=%.0f unfinished=%d first=%s evals=%s n_iter=%s loss=%s" % (
        n, os.path.basename(lib), line["all_finished"], line["value"], bad.size, bad[:8].tolist(),
        o["stats_func_evals"][bad[:4]].tolist(), o["stats_n_iter"][bad[:4]].tolist(), o["stats_final_loss"][bad[:4]].tolist()), flush=True)
    out.append(o)
for n in range(1, len(out)):
    d = np.abs(out[0]["glob"] - out[n]["glob"]).reshape(out[0]["glob"].shape[0], -1).max(1)
    print("glob 0 vs %d: max %.4g m, windows differing %d of %d; evals differ in %d" % (
        n, d.max(), int((d > 0).sum()), d.size, int((out[0]["stats_func_evals"] != out[n]["stats_func_evals"]).sum())))
