/*
 * gem_hip.h -- C ABI of the MI355X (gfx950) window optimiser for GlobalEgoMocap's hot path.
 *
 * The reference has no FFI of its own: its boundary for this path is two Python call signatures
 * (optimizer.py:311-314 `main`, optimizer.py:242-276 `optimize_pose_seq_pytorch_LBFGS`) plus the
 * checkpoint / pickle / camera-json schemas.  This header is what a Python
 * (ctypes) binding of those two calls binds to; `globalegomocap_amd/_capi.py` is that binding and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every entry point returns 0 on success, non-zero on failure; `gem_last_error()` gives the text
 *     (thread-local).  Nothing is printed, nothing aborts.
 *   - `d_*` arguments are DEVICE pointers owned by the caller (e.g. torch tensors); the library never
 *     frees or keeps them beyond the call.  `h_*` are host pointers.
 *   - `stream` is a hipStream_t passed as void*; all work of a call is enqueued on it, no call
 *     synchronises the device (except gem_create / gem_load_vae / gem_destroy, which allocate).
 *   - one handle per device; a handle is not thread-safe; no hidden RNG (eps is an input, D5).
 *   - windows are independent: `B` windows per call, each T frames x J joints (T=10, J=15).
 */
#ifndef GEM_HIP_H
#define GEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEM_MAX_HIDDEN 8
#define GEM_MAX_POLY 16
#define GEM_MAX_JOINTS 16

typedef struct gem_handle gem_handle;

/* Geometry of the path.  Replaces the constructor arguments of BodyPoseOptimizer
 * (optimizer.py:36-71), ConvVAE (networks/models/SeqConvVAE.py:11-30) and
 * FishEyeCameraCalibrated (utils/fisheye/FishEyeCalibrated.py:7-15). */
typedef struct gem_config {
    int32_t seq_len;                    /* T, frames per window (10) */
    int32_t n_joints;                   /* J (15) */
    int32_t latent_dim;                 /* D (2048) */
    int32_t n_hidden;                   /* number of encoder conv blocks (5) */
    int32_t hidden[GEM_MAX_HIDDEN];     /* encoder channel widths (64,64,128,256,512) */
    int32_t heat_h, heat_w;             /* heat-map size (64,64) */
    int32_t n_poly;                     /* number of polynomialW2C coefficients (11 or 14) */
    double  poly[GEM_MAX_POLY];         /* rho(theta) = sum poly[i] theta^i */
    double  cx, cy;                     /* image centre = intrinsic[0][2], intrinsic[1][2] */
    int32_t parents[GEM_MAX_JOINTS];    /* kinematic parents (optimizer.py:34) */
    int32_t max_windows;                /* capacity: largest B any call will pass */
    int32_t device;                     /* HIP device ordinal */
} gem_config;

/* Energy weights = BodyPoseOptimizer.set_weights (optimizer.py:73-79); gmm_weight is accepted by the
 * Python mirror and ignored exactly as the reference ignores it (D4). */
typedef struct gem_energy_weights {
    double w3d, smooth, bone, vae, reproj;
} gem_energy_weights;

/* torch.optim.LBFGS arguments as used at optimizer.py:261-262 (+ _strong_wolfe's constants). */
typedef struct gem_lbfgs_opts {
    double  lr;             /* 2 */
    int32_t max_iter;       /* 25 */
    int32_t max_eval;       /* max_iter*5/4 = 31 */
    int32_t history;        /* 100 (never reached: at most max_iter-1 pairs) */
    int32_t reserved;
    double  tol_grad;       /* 1e-7 */
    double  tol_change;     /* 1e-6 */
    double  c1, c2;         /* 1e-4, 0.9 */
    double  ls_tol_change;  /* 1e-9 */
} gem_lbfgs_opts;

typedef struct gem_window_stats {
    int32_t n_iter;         /* state['n_iter'] */
    int32_t func_evals;     /* state['func_evals'] */
    float   final_loss;
    int32_t status;         /* bit 0: finished (0 = still running: a bug); bit 1: a closure value was NaN -- a joint exactly on
                             * the optical axis, where the reference raises Exception("norm is zero!") (FishEyeCalibrated.py:124-127) */
} gem_window_stats;

enum { GEM_STAGE_LOCAL = 0, GEM_STAGE_GLOBAL = 1 };

const char* gem_last_error(void);
int gem_version(void);

/* BodyPoseOptimizer.__init__ minus checkpoint loading. */
int  gem_create(const gem_config* cfg, gem_handle** out);
void gem_destroy(gem_handle* h);

/* Two lanes: gem_optimize_windows calls of at least `min_windows` windows run as two half-batches on two streams (the caller's
 * and one owned by the handle), shifted by half an evaluation round so that the HBM-bound L-BFGS advance of one half shares the
 * device with the matrix-bound kernels of the other (windows are independent: optimizer.py:370).  Results are bitwise those of
 * one lane wherever no product is cut along K (any threshold >= 4352 windows guarantees it); outputs, statistics and
 * gem_read_trace are assembled in window order.  0 = always one lane = the DEFAULT since round 4: the bf16 tail now keeps two
 * workgroups per CU busy by itself and one lane measures faster (8192 windows: 289 k vs 247-259 k windows/s); the call stays for
 * devices / batch shapes where the split pays.  Costs a second workspace (half the size of the first) when switched on. */
int gem_set_lanes(gem_handle* h, int min_windows);

/* Arithmetic of the decoder / encoder products (every energy term is always fp32 arithmetic on the fp32 decoded pose):
 *   0 = fp32 MFMA everywhere (default, BASELINE configs[1]);
 *   1 = "bf16x3": operands of the wide products split into bf16 hi+lo, three bf16 MFMAs per product, fp32 accumulate
 *       (error ~2^-16 relative, i.e. fp32-grade, at ~1.5x the fp32 rate); the narrow tail layers stay fp32;
 *   2 = bf16 operands, fp32 accumulate (BASELINE configs[2..4] "bf16 VAE decoder / fp32 energy"): the wide products always; the
 *       narrow tail layers too, at every batch size (csrc/tail_bf16.hip: bf16 weights and bf16 activations between the layers;
 *       1..8 windows per workgroup, chosen from the batch size, every choice bitwise the same (asserted by the tests; the library
 *       is compiled with -ffp-contract=on so that no template instantiation fuses multiply-adds differently from another) -- so the
 *       tail's part of a window's bf16 result does not depend on the size of the batch it arrives in.  It is the fp32 result plus zero-mean noise of the order of 2^-9 per decoded
 *       coordinate (tests/test_hip_full_size.py::test_bf16_on_fitted_vae_against_the_oracle: ~1.3 mm per window on fitted
 *       weights, 0.06 mm on a sequence's MPJPE).
 * May be switched at any time between calls. */
enum { GEM_PRECISION_F32 = 0, GEM_PRECISION_BF16X3 = 1, GEM_PRECISION_BF16 = 2 };
int gem_set_precision(gem_handle* h, int mode);

/* network.load_state_dict(torch.load(path)['state_dict']) (optimizer.py:59-63).  `h_blobs[i]` is the
 * i-th float32 tensor of the state_dict in the order of globalegomocap_amd.vae.VAEShape.schema(),
 * `n_elem[i]` its element count (checked).  BatchNorm (eval) is folded into the convolutions and the
 * weights are re-packed for the MFMA kernels inside. */
int gem_load_vae(gem_handle* h, int stage, int n_blobs, const float* const* h_blobs, const int64_t* n_elem);

/* mean_bone_length of a chunk (optimizer.py:42-43,89-94): d_pose [n_frames,J,3] f32 -> d_out [J] f32. */
int gem_mean_bone_length(gem_handle* h, const float* d_pose, int n_frames, float* d_out, void* stream);

/* ConvVAE.get_latent_space (SeqConvVAE.py:184-189): d_pose [B,T,J*3] f32, d_eps [B,D] f32 (may be
 * NULL: z = mu).  Any of d_mu/d_logvar/d_z may be NULL. */
int gem_encode(gem_handle* h, int stage, int B, const float* d_pose, const float* d_eps,
               float* d_mu, float* d_logvar, float* d_z, void* stream);

/* ConvVAE.decode_to_bodypose (SeqConvVAE.py:131-140): d_z [B,D] -> d_pose [B,T,J,3] f32. */
int gem_decode(gem_handle* h, int stage, int B, const float* d_z, float* d_pose, void* stream);

/* total_loss + backward at fixed z (optimizer.py:226-240, 264-268), for parity tests:
 * d_pose_init [B,T,J,3] f32 (the stage's input pose), d_heat [n_frames,H,W,J] f32 (pickle layout,
 * may be NULL when reproj == 0), d_frame0 [B] int32 first frame of each window, d_mean_bone [B,J] f32.
 * Outputs (each may be NULL): d_energy [B] f64, d_parts [B,5] f64 (E_3d,E_smooth,E_bone,E_vae,E_reproj),
 * d_dz [B,D] f32, d_pose [B,T,J,3] f32 (decoded pose). */
int gem_energy_grad(gem_handle* h, int stage, int B, const float* d_z, const float* d_pose_init,
                    const float* d_heat, const int32_t* d_frame0, const float* d_mean_bone,
                    const gem_energy_weights* w, double* d_energy, double* d_parts, float* d_dz,
                    float* d_pose, void* stream);

/* BodyPoseOptimizer.optimize_pose_seq_pytorch_LBFGS for B windows at once (optimizer.py:242-276):
 * encode -> L-BFGS/strong-Wolfe over z through decoder + energies -> decode.
 * d_pose_in [B,T,J,3] f32, d_eps [B,D] f32, d_pose_out [B,T,J,3] f32, d_stats [B] (may be NULL). */
int gem_optimize_stage(gem_handle* h, int stage, int B, const float* d_pose_in, const float* d_heat,
                       const int32_t* d_frame0, const float* d_mean_bone, const float* d_eps,
                       const gem_energy_weights* w, const gem_lbfgs_opts* opt, float* d_pose_out,
                       gem_window_stats* d_stats, void* stream);

/* Closure values of the LAST stage run on this handle (the `total_loss` values torch.optim.LBFGS.step's closure returned,
 * optimizer.py:263-268): d_out [n_rounds][B] f64, row r = the value window b's optimiser consumed in evaluation round r,
 * NaN once the window had finished (a window's evaluations are rounds 0..func_evals-1).  n_rounds <= 64.  For parity
 * tests against the reference's closure traces; after gem_optimize_windows it holds the global stage's values. */
int gem_read_trace(gem_handle* h, int B, int n_rounds, double* d_out, void* stream);

/* The window loop body of main() (optimizer.py:370-423) for B windows at once: local stage,
 * relative-global transform X_rel[t] = C0^-1 C_t X_loc[t] in float64 (utils/utils.py:99-112), global
 * stage, X_glob = C0 X_rel (optimizer.py:302-308).
 *   d_local_pose [n_frames,J,3] f32   estimated_local_skeleton, frames stored once
 *   d_cams       [n_frames,4,4] f64   camera_pose_list
 *   d_heat       [n_frames,H,W,J] f32 heatmap_list
 *   d_frame0     [B] int32            first frame of each window
 *   d_mean_bone  [B,J] f32
 *   d_eps_local / d_eps_global [B,D] f32
 * Outputs: d_mid_local [B,T,J,3] f32 (stage-A result, may be NULL), d_global [B,T,J,3] f64
 * (refined global pose), d_stats [2*B] (local stats then global stats, may be NULL). */
int gem_optimize_windows(gem_handle* h, int B, const float* d_local_pose, const double* d_cams,
                         const float* d_heat, const int32_t* d_frame0, const float* d_mean_bone,
                         const float* d_eps_local, const float* d_eps_global,
                         const gem_energy_weights* w_local, const gem_energy_weights* w_global,
                         const gem_lbfgs_opts* opt, float* d_mid_local, double* d_global,
                         gem_window_stats* d_stats, void* stream);

/* The reprojection term (optimizer.py:139-149) keeps the four heat-map texels under every joint from one evaluation of a
 * stage to the next and re-reads them as one record while the joint stays inside the same texel block (same values, bit for
 * bit; fewer scattered cache lines).  On by default; this switch exists so that a test can prove "bit for bit". */
int gem_set_texel_cache(gem_handle* h, int on);

/* hipGraph replay of whole optimisation calls (BASELINE configs[4] "hipGraph-captured inner step"; the loop being captured
 * replaces optimizer.py:261-270 for every window of the call).  With graphs on, gem_optimize_stage / gem_optimize_windows
 * run eagerly the first time they see a given signature (batch size, precision, every pointer argument, weights, options,
 * stream), capture the second call into a hipGraph and replay that graph from then on: one hipGraphLaunch instead of ~700
 * kernel launches per call.  Results are bitwise those of the eager path.  Needs a non-default stream (the legacy default
 * stream cannot be captured: such calls stay eager) and profiling off.  The caller must pass the SAME buffers to get
 * replays; a call with other pointers is a new signature (at most 16 are cached per handle).
 * A captured call holds the ADDRESSES of the caller's buffers.  A caller that frees buffers a captured call used must call
 * gem_graph_enable(h, 0) first -- it synchronises the device and drops every cached graph (gem_graph_enable(h, 1) switches replay
 * on again) -- so that no graph can be replayed on whatever is allocated at those addresses next (ROCm 7.2: such a replay ended in
 * a GPU memory access fault even with equally sized new buffers at the same addresses).
 * gem_graph_stats: number of captures and replays so far. */
int gem_graph_enable(gem_handle* h, int on);
int gem_graph_stats(gem_handle* h, int64_t* n_captures, int64_t* n_replays);

/* ---- sequence post-processing: the reporting side of the path, on the same stream ----
 * Both calls may grow an internal scratch buffer (hipMalloc, synchronous) the first time a larger sequence is seen. */

/* merge_batches (optimizer.py:425-437) per chunk, then optionally gaussian_filter1d(sigma=1, axis=0) per chunk
 * (optimizer.py:448-450).  d_windows [n_chunks*windows_per_chunk, T, J, 3] f64 (e.g. d_global of
 * gem_optimize_windows) -> d_out [n_chunks*frames_per_chunk, J, 3] f64 with
 * frames_per_chunk = windows_per_chunk*(T-overlap)+overlap. */
int gem_merge_windows(gem_handle* h, const double* d_windows, int n_chunks, int windows_per_chunk, int overlap,
                      int smooth, double* d_out, void* stream);

/* calculate_errors (calculate_errors.py:114-179): d_est / d_mid / d_opt / d_gt [n_frames,J,3] f64 (metres),
 * h_bone_mm [J] reference bone lengths in mm (utils/skeleton.py:102-110).  d_out [17+J] f64 in the order of
 * the reference's result dict: original/mid/optimized_global_mpjpe, original/optimized_camera_pos_error,
 * original/mid/optimized aligned camera error, original/mid/optimized sequence-aligned mpjpe,
 * per-frame-Procrustes mpjpe x3, bone-length-normalised mpjpe x3, joints_error[J].
 * Includes the batched 3x3 SVD Umeyama (utils/rigid_transform_with_scale.py:18-43) and the skeleton
 * re-growth (utils/skeleton.py:124-136).  Needs n_joints >= 12 (hip joints 7 and 11). */
int gem_calculate_errors(gem_handle* h, const double* d_est, const double* d_mid, const double* d_opt,
                         const double* d_gt, int n_frames, const double* h_bone_mm, double* d_out, void* stream);

/* gem_calculate_errors for each of n_chunks equally long sequences laid end to end (d_* [n_chunks*frames_per_chunk,J,3] f64) in ONE
 * call: the per-chunk `main()` reports that optimize_whole_sequence.py:55-104 collects.  d_out [n_chunks][17+J] f64. */
int gem_calculate_errors_chunks(gem_handle* h, const double* d_est, const double* d_mid, const double* d_opt, const double* d_gt,
                                int n_chunks, int frames_per_chunk, const double* h_bone_mm, double* d_out, void* stream);

/* ---- input lifting: raw network outputs -> estimated_local_skeleton ----
 * Skeleton.set_skeleton_from_file + set_skeleton + get_max_preds (utils/skeleton.py:74-90,32-45,176-204) followed by
 * FishEyeCameraCalibrated.camera2world (utils/fisheye/FishEyeCalibrated.py:18-33), without the bone-length resize
 * (the reference's data preparation passes bone_length_file=None, MakeDataForOptimization/process_test_data.py:59-62).
 *   d_heat   [n_frames,H,W,J] f32   heat-maps as stored in the .mat files / the pickle
 *   d_depth  [n_frames,J] f64       predicted joint distances
 *   h_poly_c2w                      the calibration's polynomialC2W (ascending powers)
 *   upscale, pad_x, pad_y           16, 128, 0: cv2.resize(64 -> 1024, INTER_NEAREST) + np.pad 128 columns each side
 * Outputs (either may be NULL): d_out64 [n_frames,J,3] f64 (what the reference pickles), d_out32 the same as f32
 * (what gem_optimize_windows consumes).  Family 3 of the profiling hook times this kernel (bytes instead of flops). */
int gem_lift_skeleton(gem_handle* h, const float* d_heat, const double* d_depth, int n_frames, const double* h_poly_c2w,
                      int n_poly_c2w, int upscale, int pad_x, int pad_y, double* d_out64, float* d_out32, void* stream);

/* ---- chunk files: `<chunk>/test_data.pkl` as the reference writes and reads it ----
 * The file is a pickled dict of lists of numpy arrays (written at MakeDataForOptimization/process_test_data.py:149-157, read at
 * optimizer.py:315-324); its heat-maps are what scipy.io.loadmat returned (process_test_data.py:65-67): [H,W,J] arrays in
 * FORTRAN order, float32 or float64.  These calls take such a file to the device without building a Python object per array.
 * None of them needs a gem_handle; the host-side ones need no GPU. */
enum { GEM_DT_F32 = 0, GEM_DT_F64 = 1 };
enum { GEM_PICKLE_UNSUPPORTED = 2 };     /* return code: a well-formed call on a file outside the subset -- un-pickle it the ordinary way */

typedef struct gem_pickle_array {
    int64_t offset;         /* of the array's raw data, in bytes from the start of the file */
    int64_t nbytes;
    int32_t dtype;          /* GEM_DT_F32 / GEM_DT_F64 (little-endian) */
    int32_t ndim;           /* 0..4 */
    int32_t fortran;        /* 1: the data are in Fortran order (ndarray.flags.f_contiguous and not c_contiguous) */
    int32_t key;            /* index into the `keys` the scan was given */
    int64_t shape[4];       /* unused trailing dimensions are 1 */
} gem_pickle_array;

/* Interprets the pickle in h_image[0, len) (protocols 2-5; the opcodes a dict of lists of ndarrays consists of) WITHOUT constructing
 * or calling anything it names, and reports where the arrays of the dict's entries `keys[0 .. n_keys)` lie: `out` receives the
 * arrays key by key in list order (at most `cap`), counts[k] the number of arrays under keys[k], or -1 when the dict has no such
 * key.  Returns GEM_PICKLE_UNSUPPORTED (text in gem_last_error) for anything else: other opcodes, an entry that is not a list of
 * float32 / float64 ndarrays, protocol-2 arrays (their bytes are stored as text), big-endian data. */
int gem_pickle_scan(const void* h_image, int64_t len, const char* const* keys, int n_keys, gem_pickle_array* out, int64_t cap,
                    int64_t* counts);

/* n equally shaped arrays found by gem_pickle_scan -> h_out [n][shape] float64, C order (np.asarray(list_of_arrays) as at
 * optimizer.py:318-323 for the skeleton / camera lists). */
int gem_pickle_gather_f64(const void* h_image, int64_t len, const gem_pickle_array* arrays, int64_t n, double* h_out);

/* Heat-maps out of a DEVICE image of (part of) the file: d_image[0, image_len) holds file bytes, d_offsets [n] the byte position of
 * every frame's raw data in it (any alignment); dtype / fortran as gem_pickle_scan reported them (all frames alike).
 * d_out [n, heat_h, heat_w, n_joints] f32 = what `torch.from_numpy(np.asarray(heatmap_list)).float()` holds (optimizer.py:324,248):
 * the Fortran order undone, float64 rounded to nearest even.  d_image must be 4-byte aligned and readable up to image_len rounded
 * UP to a multiple of 4; d_out 16-byte aligned; n <= 65535. */
int gem_heat_gather(const void* d_image, int64_t image_len, const int64_t* d_offsets, int64_t n, int heat_h, int heat_w,
                    int n_joints, int dtype, int fortran, float* d_out, void* stream);

/* One chunk file as an object: gem_chunk_open opens and maps `path`, scans it for `keys` (gem_pickle_scan) and keeps the table.
 * gem_chunk_count: arrays under keys[key] (-1: no such key); gem_chunk_bytes: size of the file.
 * gem_chunk_info: info[8] = { count, ndim, dtype, fortran, shape[0..3] } of the arrays under keys[key]; ndim = -1 when they are not
 * all of one shape, type and order.  gem_chunk_offsets: the byte offset of every array's raw data in the file (h_out [count]).
 * gem_chunk_gather_f64: gem_pickle_gather_f64 on the key's arrays (n_out = capacity of h_out in doubles). */
typedef struct gem_chunk gem_chunk;
int     gem_chunk_open(const char* path, const char* const* keys, int n_keys, gem_chunk** out);
void    gem_chunk_close(gem_chunk* c);
int64_t gem_chunk_bytes(const gem_chunk* c);
int64_t gem_chunk_count(const gem_chunk* c, int key);
int     gem_chunk_info(const gem_chunk* c, int key, int64_t* info);
int     gem_chunk_offsets(const gem_chunk* c, int key, int64_t* h_out, int64_t cap);
int     gem_chunk_gather_f64(const gem_chunk* c, int key, double* h_out, int64_t n_out);

/* A file on its way to the device: `path` is read (pread, `slice_bytes` at a time) into the caller's PINNED buffer h_pinned and sent
 * on to d_image slice by slice (hipMemcpyAsync on `stream` of `device`: the next slice is read while the last one crosses PCIe), so
 * that d_image[0, *file_bytes) is an image of the file once the stream has passed the call's work -- what gem_heat_gather reads, with
 * the offsets gem_chunk_offsets reports.  Both buffers must hold buffer_bytes >= file size + 8; h_pinned must stay untouched until the
 * stream has passed.  Nothing synchronises; needs no scan of the file, so it can run beside gem_chunk_open on another thread.
 * Calls on different files may run concurrently from different threads (one stream and one pair of buffers each). */
int gem_file_stage(const char* path, int device, void* h_pinned, void* d_image, int64_t buffer_bytes, int64_t slice_bytes,
                   int64_t* file_bytes, void* stream);

/* Timing hook for bench.py's roofline: average device time (ms) of the launches of the dominant
 * kernel family since the last reset, measured with HIP events on the launch stream.
 * family: 0 = decoder_input GEMMs (forward + backward-data), 1 = fused tail / energy kernel, 2 = L-BFGS advance,
 *         3 = input lifting (`flops` then returns algorithmic bytes).
 * `gem_profile_enable(h, 1)` turns event recording on (off by default: events cost launches). */
int gem_profile_enable(gem_handle* h, int on);
int gem_profile_read(gem_handle* h, int family, double* total_ms, int64_t* n_launches, double* flops);
/* Names of the kernels launched for `family` while event recording was on, since the last call (a "; "-separated list of
 * demangled names without parameter lists, i.e. as rocprofv3 --kernel-trace --stats prints them), written to buf (truncated to
 * buf_len - 1 characters).  bench.py labels its roofline objects with this, so that the label is the kernel that actually ran. */
int gem_profile_kernels(gem_handle* h, int family, char* buf, int buf_len);

/* ---- VAE training on the device: the loop body of networks/train.py:65-108 ----
 * One gem_trainer_step = model.train() forward (BatchNorm1d batch statistics; running statistics updated with `bn_momentum`,
 * unbiased variance), ConvVAE.loss_function (networks/models/SeqConvVAE.py:191-219: F.mse_loss(recons, input) + kld_weight *
 * mean_b(-0.5 * sum_d(1 + logvar - mu^2 - exp(logvar))), `kld_weight` being train.py:89's M_N), loss.backward() and -- when
 * `update` is non-zero -- one torch.optim.Adam step (train.py:60: lr, betas, eps, L2 weight_decay added to the gradient).
 * update = 1 leaves every gradient in the gradient arena (p.grad after the step).  update = 2 is the training loop's mode: the two
 * linear layers (fc_mu | fc_var, decoder_input: 97 % of the parameters) form their weight gradient INSIDE their Adam step and do
 * not write it to the arena (afterwards gem_trainer_download(what = 1) returns NaN in those two ranges, and gem_trainer_arena(what = 1)
 * / gem_trainer_apply fail until a step with update = 0 or 1 has filled the arena again); parameters, moments, statistics and losses are the same
 * up to summation order (at batches of at most 64 windows the same pass over the weights also forms the layers' backward-data
 * products; what Adam's eps makes of last-bit differences: tests/test_hip_train.py::test_training_loop_mode_steps_like_the_default_mode).
 *
 * Parameters, gradients and both Adam moments are fp32 arenas of `n_params` floats in the packed device layout; running
 * statistics an arena of `n_stats` floats.  Arena order (every width padded to a multiple of 64, padding zero):
 *   encoder block i:  W [3][N][K] (Conv1d weight[n][k][tap] at [tap][n][k]), bias [N], BN gamma [N], BN beta [N]
 *   fc_mu | fc_var:   W [2*Dp][T*topp] (row d = fc_mu d, row Dp + d = fc_var d; column t*topp + c = flattened feature c*T + t),
 *                     bias [2*Dp]
 *   decoder_input:    W [T*topp][Dp] (row t*topp + c = output feature c*T + t), bias [T*topp]
 *   decoder block i:  W [3][N][K] = the equivalent Conv1d taps of ConvTranspose1d(k=3,s=1,p=1): weight[k][n][2-tap] at
 *                     [tap][n][k]; bias, gamma, beta (the final Conv1d has no BatchNorm: W, bias only)
 *   statistics:       per BatchNorm layer in the same order: running_mean [N], running_var [N]
 * globalegomocap_amd/vae_train.py converts between this and the checkpoint schema of the reference.
 *
 *   d_pose    [B,T,3J] f32 device   the batch (train.py:82 after the float() cast)
 *   d_eps     [B,D]    f32 device   the reparameterisation noise (torch.randn_like at SeqConvVAE.py:167)
 *   d_losses  [3]      f64 device   loss, Reconstruction_Loss, KLD (may be NULL)
 * 2 <= B <= cfg.max_windows; update must be 0, 1 or 2.  Everything is enqueued on `stream`; nothing synchronises. */
typedef struct gem_trainer gem_trainer;
typedef struct gem_train_opts {
    double  lr, beta1, beta2, eps, weight_decay;   /* torch.optim.Adam: 1e-3, 0.9, 0.999, 1e-8, train.py's --weight_decay */
    double  kld_weight;                            /* M_N = kl_weight * batch / len(dataset) (train.py:89) */
    double  bn_momentum;                           /* 0.1 */
    int32_t recon_sum;                             /* 0: F.mse_loss mean (the reference); 1: summed squared error */
    int32_t reserved;
} gem_train_opts;
int  gem_trainer_create(const gem_config* cfg, gem_trainer** out);
void gem_trainer_destroy(gem_trainer* t);
int  gem_trainer_sizes(gem_trainer* t, int64_t* n_params, int64_t* n_stats);
/* what: 0 parameters (resets the Adam step count), 1 gradients (download only), 2 running statistics, 3 / 4 Adam moments */
int  gem_trainer_upload(gem_trainer* t, int what, const float* h_src, int64_t n);
int  gem_trainer_download(gem_trainer* t, int what, float* h_dst, int64_t n);
int  gem_trainer_set_step(gem_trainer* t, int64_t step);
int  gem_trainer_step(gem_trainer* t, int B, const float* d_pose, const float* d_eps, const gem_train_opts* opts, int update,
                      double* d_losses, void* stream);
/* Data-parallel training (one process per GPU, every rank its own batch; the reference trains on one device, networks/train.py:58):
 * gem_trainer_step(..., update = 0, ...) leaves the rank's gradients in the gradient arena; the caller all-reduces that ONE flat
 * buffer (gem_trainer_arena returns the device address of arena `what` and its length in floats; RCCL through torch.distributed
 * in vae_train.py) and applies the Adam step with gem_trainer_apply, which scales the gradients by grad_scale (1 / world size)
 * on the fly.  BatchNorm statistics stay per rank, as under torch's DistributedDataParallel without SyncBatchNorm. */
int  gem_trainer_arena(gem_trainer* t, int what, void** d_ptr, int64_t* n);
int  gem_trainer_apply(gem_trainer* t, const gem_train_opts* opts, double grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEM_HIP_H */
