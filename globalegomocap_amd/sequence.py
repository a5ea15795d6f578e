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


def merge_chunks(windows, n_chunks, overlap=OVERLAP):
    """merge_batches for n_chunks chunks of equally many windows at once: [n_chunks*wpc, T, ...] -> [n_chunks, fpc, ...]
    (same arithmetic per element as merge_batches, vectorised over the chunks)."""
    w = np.asarray(windows)
    wpc, T = w.shape[0] // n_chunks, w.shape[1]
    w = w.reshape((n_chunks, wpc, T) + w.shape[2:])
    if overlap == 0:
        return w.reshape((n_chunks, wpc * T) + w.shape[3:])
    step = T - overlap
    out = np.empty((n_chunks, wpc * step + overlap) + w.shape[3:], dtype=w.dtype)
    out[:, :step] = w[:, 0, :step]
    for i in range(1, wpc):
        out[:, i * step:i * step + overlap] = (w[:, i - 1, -overlap:] + w[:, i, :overlap]) / 2
        out[:, i * step + overlap:(i + 1) * step] = w[:, i, overlap:step]
    out[:, wpc * step:] = w[:, -1, -overlap:]
    return out


def final_smooth(seq):
    """scipy gaussian_filter1d(sigma=1, axis=0) (optimizer.py:448-450)."""
    from scipy.ndimage import gaussian_filter1d
    return gaussian_filter1d(np.asarray(seq), sigma=1, axis=0)


def relative_global_numpy(local, cams):
    """float64 host twin of the device transform, for the *unoptimised* sequences that main() also
    returns (utils/utils.py:99-112): X_rel[t] = C0^-1 C_t X_loc[t], per window."""
    local = np.asarray(local, dtype=np.float64)
    cams = np.asarray(cams, dtype=np.float64)
    M = np.matmul(np.linalg.inv(cams[:, 0])[:, None], cams)
    return np.matmul(local, np.swapaxes(M[..., :3, :3], -1, -2)) + M[..., None, :3, 3]


def to_global_numpy(rel, cams):
    """X_glob = C0 X_rel (optimizer.py:302-308), per window, float64."""
    rel = np.asarray(rel, dtype=np.float64)
    c0 = np.asarray(cams, dtype=np.float64)[:, 0]
    return np.matmul(rel, np.swapaxes(c0[:, None, :3, :3], -1, -2)) + c0[:, None, None, :3, 3]
