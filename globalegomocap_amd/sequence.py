"""Sliding-window bookkeeping of the sequence pipeline (host side, numpy).

Window starts, overlap averaging and the final Gaussian smoothing follow the reference's `main()`
(`optimizer.py:370,425-450`); the window optimisation itself is one batched device call
(`WindowEngine.optimize_windows`).
"""
import numpy as np

SEQ_LEN = 10       # optimizer.py:328
OVERLAP = 2        # optimizer.py:330


def window_starts(n_frames, seq_len=SEQ_LEN, overlap=OVERLAP):
    """range(0, N - seq_len + 1, seq_len - overlap)  (optimizer.py:370); the last N-2.. frames of a
    100-frame chunk are never optimised (D7)."""
    return np.arange(0, n_frames - seq_len + 1, seq_len - overlap, dtype=np.int32)


def cut_windows(frames, starts, seq_len=SEQ_LEN):
    frames = np.asarray(frames)
    return np.stack([frames[s:s + seq_len] for s in starts]) if len(starts) else frames[:0].reshape((0, seq_len) + frames.shape[1:])


def merge_batches(windows, overlap=OVERLAP):
    """Average the `overlap` frames shared by neighbouring windows (optimizer.py:425-437)."""
    w = np.asarray(windows)
    if overlap == 0:
        return np.concatenate(w)
    parts = [w[0][:-overlap]]
    for i in range(len(w) - 1):
        parts.append((w[i][-overlap:] + w[i + 1][:overlap]) / 2)
        parts.append(w[i + 1][overlap:-overlap])
    parts.append(w[-1][-overlap:])
    return np.concatenate(parts)


def final_smooth(seq):
    """scipy gaussian_filter1d(sigma=1, axis=0) (optimizer.py:448-450)."""
    from scipy.ndimage import gaussian_filter1d
    return gaussian_filter1d(np.asarray(seq), sigma=1, axis=0)


def relative_global_numpy(local, cams):
    """float64 host twin of the device transform, for the *unoptimised* sequences that main() also
    returns (utils/utils.py:99-112): X_rel[t] = C0^-1 C_t X_loc[t], per window."""
    local = np.asarray(local, dtype=np.float64)
    cams = np.asarray(cams, dtype=np.float64)
    M = np.einsum("bij,btjk->btik", np.linalg.inv(cams[:, 0]), cams)
    return np.einsum("btij,btnj->btni", M[..., :3, :3], local) + M[..., None, :3, 3]


def to_global_numpy(rel, cams):
    """X_glob = C0 X_rel (optimizer.py:302-308), per window, float64."""
    rel = np.asarray(rel, dtype=np.float64)
    c0 = np.asarray(cams, dtype=np.float64)[:, 0]
    return np.einsum("bij,btnj->btni", c0[:, :3, :3], rel) + c0[:, None, None, :3, 3]
