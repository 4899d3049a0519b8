/*
 * nerftex.h -- C ABI of libnerftex_hip.so: the MI355X (gfx950) implementation of NeRF-Tex's
 * volumetric render path.
 *
 * This is the drop-in boundary of the project.  The reference
 * (hbaatz/nerf-tex) has no FFI of its own on this path -- it is eager TensorFlow called through
 * Python objects -- so the entry points below are what a binding for each reference function would
 * bind.  Each one cites the reference interface it replaces (file:line in /root/reference).
 *
 * Conventions
 *   - plain C, `extern "C"`, no torch / TF / C++ types in any signature;
 *   - every `float*` marked DEVICE is a caller-owned device pointer (row-major float32, last
 *     dimension fastest, exactly the layouts of the reference tensors); HOST pointers are read
 *     synchronously during the call;
 *   - the library owns only what `ntx_create` / `ntx_reserve` / `ntx_comm_create` / `ntx_instancer_create` /
 *     `ntx_instancer_reserve` / `ntx_instancer_set_mesh` allocate (the packed weight image, the hit-list scratch, the communicator,
 *     the instance matrices and the instancer's hit lists); no other entry point allocates or frees device memory;
 *   - every entry point is asynchronous on `stream` (a `hipStream_t` passed as `void*`; NULL = the
 *     null stream) and returns an `ntx_status` (0 = ok, negative = error).  The message of the
 *     last error on the calling thread is returned by `ntx_last_error()`;
 *   - a context belongs to one device; calls on one context are not re-entrant.  Multi-GPU =
 *     one context (and one process) per device.
 */
#ifndef NERFTEX_H
#define NERFTEX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTX_ABI_VERSION 7

typedef struct ntx_ctx ntx_ctx;
typedef void *ntx_stream; /* hipStream_t */

typedef enum ntx_status {
    NTX_OK = 0,
    NTX_E_INVALID = -1,     /* bad argument (NULL pointer, negative size, n_samples < 2 ...) */
    NTX_E_UNSUPPORTED = -2, /* model architecture outside what the HIP kernels are built for */
    NTX_E_HIP = -3,         /* a HIP runtime call failed; ntx_last_error() carries hipGetErrorString */
    NTX_E_NODEVICE = -4     /* no usable gfx950 device */
} ntx_status;

/* Model architecture = the kwargs of network.model.ParamNerf (model.py:58) / Nerf (model.py:9). */
typedef enum ntx_model_kind {
    NTX_MODEL_PARAMNERF = 0,
    NTX_MODEL_NERF = 1,
    NTX_MODEL_PARAMNERF_EX = 2   /* a ParamNerf whose descriptor is the extended struct ntx_model_desc_ex below: param_depth > 0 */
} ntx_model_kind;
/* layer.FourierFeatures (layer.py:8-23) / layer.IntegratedPositionalEncoding (layer.py:25-41) */
typedef enum ntx_pos_encoding { NTX_POS_FOURIER = 0, NTX_POS_IPE = 1 } ntx_pos_encoding;

/* Built: ParamNerf with n_parameters = [g, a], g <= 4, a <= 8 (tuned kernel families for the shipped configs [1,6], [1,4], [2,3];
 * every other combination runs on one generic family whose rows for the absent parameters are zero, ~2 % more matrix work),
 * plain Nerf, and one IntegratedPositionalEncoding family [1,3] -- these at the architecture every reference config uses: depth 8,
 * width 256, skips [4], color_depth 1.  Any OTHER architecture of model.py:58 / :9 with FourierFeatures embeddings -- depth 1..24,
 * width 2..256, color_depth 0..4, any skips below depth-1 -- runs on the "flex" family: the same MFMA segments in a loop over layers
 * (narrower layers zero-padded to 256), float32 only (NTX_FLAG_FP16X3: NTX_E_UNSUPPORTED), nothing hoisted per ray; param_depth 1..4
 * through the extended descriptor below (ntx_model_desc_ex).  n_freq_bands (layer.py:11) up to 10 / 4 / 4 for position / direction /
 * parameters, every family and precision: the kernels evaluate their 10 / 4 / 4 bands, the ones a model does not have meet zero weight
 * rows (exact).  No embedding_config.  Anything else: NTX_E_UNSUPPORTED. */
#define NTX_SKIP_MASK 0x40000000   /* ntx_model_desc.skip = NTX_SKIP_MASK | mask: several skip layers (bit i: i in skips) */
typedef struct ntx_model_desc {
    int32_t kind;        /* ntx_model_kind */
    int32_t n_geo;       /* n_parameters[0]: parameters concatenated to the position embedding (model.py:88-93) */
    int32_t n_app;       /* n_parameters[1]: parameters concatenated to the direction embedding (model.py:96-101) */
    int32_t n_pos;       /* 3 */
    int32_t pos_freq;    /* pos_embedding.n_freq_bands   (layer.py:11): 0..10 */
    int32_t dir_freq;    /* dir_embedding.n_freq_bands: 0..4 */
    int32_t param_freq;  /* param_embedding.n_freq_bands: 0..4 */
    int32_t depth;       /* 8; flex family: 1..24 */
    int32_t width;       /* 256; flex family: 2..256 */
    int32_t skip;        /* `skips`: the index of the single skip layer, 4 (model.py:107-108); -1 = none; or NTX_SKIP_MASK | (bit i set for
                          * every i in skips) */
    int32_t color_depth; /* ParamNerf: 1 (model.py:118), flex family: 0..4; ignored for Nerf */
    int32_t pos_encoding;/* ntx_pos_encoding of pos_embedding: FourierFeatures (n_pos 3) or IntegratedPositionalEncoding (n_pos 6) */
} ntx_model_desc;

/* kind == NTX_MODEL_PARAMNERF_EX: the pointer every entry point takes as `const ntx_model_desc *` points to one of these.
 * param_depth Dense(param_width, relu) layers shape the Fourier features of the geometry parameters and, separately, of the
 * appearance parameters before they are concatenated to pos_map / dir_map (model.py:88-101).  Built: param_depth 1..4,
 * param_width 2..128, on the flex family's float32 kernels (any depth / width / skips / color_depth it takes).  param_depth = 0
 * is the plain ParamNerf. */
typedef struct ntx_model_desc_ex {
    ntx_model_desc base;     /* base.kind = NTX_MODEL_PARAMNERF_EX */
    int32_t param_depth;
    int32_t param_width;
    int32_t reserved[6];     /* 0 */
} ntx_model_desc_ex;

/* flags of ntx_composite / ntx_render_rays */
#define NTX_FLAG_MAP_EXR 1u         /* renderer.py:182-184: colour = elu(raw)+1 instead of sigmoid  */
#define NTX_FLAG_COMPOSITE_BKGD 2u  /* renderer.py:210-211 and 85-86: add (1-A)*bkgd; culled rays = bkgd */
#define NTX_FLAG_CHECK_NUMERICS 4u  /* renderer.py:140-141: set *status_flag |= 1 on NaN/Inf outputs */
#define NTX_FLAG_FP16X3 8u          /* this call uses NTX_PRECISION_FP16X3 (below) instead of float32 Dense layers */
#define NTX_FLAG_PERTURB 16u        /* ntx_render_rays: stratified jitter of the sample depths (renderer.py:106-111), see there */
#define NTX_FLAG_RAW_NOISE 32u      /* ABI v3: sigma += N(0, raw_noise_std) per sample before the relu (renderer.py:190-192, 335-337); ntx_render_opts */

/* ABI v3: optional extras of ntx_render_rays / ntx_render_instanced / ntx_sample_depths / ntx_sample_pdf (HOST struct, read during
 * the call; NULL = all defaults).  `size` = sizeof(ntx_render_opts) of the caller's header, so that later versions can append.
 *   raw_noise_std  with NTX_FLAG_RAW_NOISE: the density regulariser of map_model_output (renderer.py:190-192; InstanceRenderer
 *                  :335-337).  The draw for (ray, sample) is the first output of tf.random.normal's Box-Muller transform
 *                  (u1 clamped to 1e-7; sqrt(-2 ln u1) sin(2 pi u2)) of words 0 and 1 of Philox4x32-10 at counter
 *                  (sample index, ray index lo, ray index hi, 1) under the call's seed: a pure function of (seed, ray, sample) like
 *                  the jitter (whose counter ends in 0).  TensorFlow's own stream cannot be reproduced; the distribution is the same.
 *   noise_seed     the seed of ntx_render_instanced, which has no perturb_seed (the others key the noise with perturb_seed).
 *   ray_index0, ray_run_length, ray_run_stride
 *                  the index that keys both generators: local ray k of the call counts as ray
 *                      ray_index0 + (k / ray_run_length) * ray_run_stride + k % ray_run_length
 *                  -- the pixel-set arithmetic of ntx_generate_rays_strided.  All zero = k itself.  A caller that renders an
 *                  image in chunks passes the chunk's first ray as ray_index0; rank r of a shard map passes its pixel set
 *                  (r * run_length, run_length, n_ranks * run_length): the image then does not depend on how it was split. */
typedef struct ntx_render_opts {
    uint32_t size;                       /* sizeof(ntx_render_opts) of the caller's header: >= NTX_RENDER_OPTS_V3_SIZE; fields beyond `size` read as 0 */
    float raw_noise_std;
    uint64_t noise_seed;
    int64_t ray_index0, ray_run_length, ray_run_stride;
    uint32_t flags;                      /* ABI v6: NTX_OPT_* */
    uint32_t reserved;
} ntx_render_opts;
#define NTX_RENDER_OPTS_V3_SIZE 40u
/* ntx_instancer_model_input: the rows of its [N,S,...] outputs behind a ray's last marching step -- dists == 0 there, and it is written in
 * full -- are left UNWRITTEN in the other six (rays_d_map, pts, t, alpha_weight, instance_id, params_map) instead of holding the defaults
 * of instancer.pyx:41-50.  For callers that hand the buffers to ntx_render_instanced, which reads a row of those only where dists > 0
 * (renderer.py:284-288): three quarters of the bytes of a carpet frame are such defaults. */
#define NTX_OPT_INSTANCER_SPARSE 1u

/* Arithmetic of the Dense layers inside ntx_render_rays, ntx_render_instanced and ntx_mlp_forward (everything else -- encoders, heads,
 * compositing -- is float32 either way).  The reference computes in float32 (TensorFlow's default dtype, model.py:104-123):
 *   NTX_PRECISION_F32    float32 matrix cores (v_mfma_f32_32x32x2_f32), the default; 1.7e-6 from the float32 reference.
 *   NTX_PRECISION_FP16X3 opt-in: weights and activations split as v = hi + lo (two IEEE halves, round to nearest even,
 *                        subnormals kept) and multiplied as hi*hi + hi*lo + lo*hi on the 16-bit matrix cores
 *                        (v_mfma_f32_32x32x16_f16) with float32 accumulation.  hi + lo carries 22 mantissa bits and a
 *                        product of two halves is exact in float32, so only the lo*lo term and the rounding of lo are lost,
 *                        ~2^-22 relative per product: 2.8e-6 from the float32 kernel on the bench image, ~2.8x faster.  Not
 *                        bit-identical to NTX_PRECISION_F32.  Range: |activation| and |weight| <= 65504, beyond that the
 *                        sample becomes inf/NaN (reported through NTX_FLAG_CHECK_NUMERICS).  Every family and all three
 *                        entry points. */
typedef enum ntx_precision { NTX_PRECISION_F32 = 0, NTX_PRECISION_FP16X3 = 1 } ntx_precision;
/* The precision is chosen PER CALL with NTX_FLAG_FP16X3 in `flags`; a context holds both weight images and no mutable
 * precision state, so two callers may share a context on different precisions. */

int ntx_abi_version(void);
const char *ntx_last_error(void);

/* Number of float32 in the reference-layout weight blob of `desc`:
 *   np.concatenate([w.ravel() for w in keras_model.get_weights()])
 * i.e. per Dense layer kernel[in,out] (row-major) then bias[out], in the order of `keras_model.layers`.  A functional
 * tf.keras.Model sorts its layers by graph depth (ties: traversal order from outputs=[color, alpha], model.py:125), NOT
 * by creation, so the order is
 *   trunk 0..7 (model.py:105-106) | feature (:114) | colour layers (:118-119, ParamNerf) | colour half (:122) | color (:123) | alpha (:111)
 * -- the alpha head comes LAST although it is created before the feature layer (the same order as the
 * `layer_with_weights-k` keys of the checkpoints logger.py:30-39 writes).  Every layer has a distinct position, so a blob in
 * another order has the right size and cannot be detected here; the Python mirror's `set_weights(list)` checks shapes.
 * Returns 0 if `desc` is not supported. */
size_t ntx_weight_count(const ntx_model_desc *desc);

/* Replaces: building the tf.keras.Model (model.py:125) + tf.train.Checkpoint restore (logger.py:33-39).
 * Packs the HOST blob into the MFMA operand layout and uploads it to `device`. */
int ntx_create(const ntx_model_desc *desc, const float *weights_host, size_t n_floats, int device,
               ntx_ctx **out);
/* Re-pack and re-upload new weights into an existing context (synchronous). */
int ntx_set_weights(ntx_ctx *ctx, const float *weights_host, size_t n_floats);
int ntx_destroy(ntx_ctx *ctx);

/* Setup-time sizing of the context's device scratch (synchronous; frees and re-allocates): after ntx_reserve(ctx, n),
 * ntx_render_rays / ntx_render_instanced accept up to n rays per call and allocate nothing.  Scratch = 8 B per ray (the
 * compacted list of rays that hit the proxy, renderer.py:58-67; the cost-ordered ray list of the instanced kernels).
 * ntx_create reserves NTX_DEFAULT_MAX_RAYS; a call with more rays than reserved fails with NTX_E_INVALID and renders
 * nothing. */
#define NTX_DEFAULT_MAX_RAYS (1 << 20)
int ntx_reserve(ntx_ctx *ctx, int64_t max_rays);

/* Replaces pixel_sampler.Full (pixel_sampler.py:14-15) + ray_sampler.rays_from_camera
 * (ray_sampler.py:39-48) + ray_sampler.Proxy (32-37) with proxy.AABB (proxy.py:13-35) [mode 0]
 * or ray_sampler.Frustum (15-21) [mode 1], for pixels [pixel0, pixel0+n_pixels) of the row-major
 * H x W grid.  c2w: HOST float[16] row-major 4x4.  b0/b1: HOST float[3] (mode 0).
 * Outputs (DEVICE): rays_o[n,3], rays_d[n,3], t[n,2], cone_scale[n,1]. */
int ntx_generate_rays(const float *c2w, int height, int width, float focal, int64_t pixel0,
                      int64_t n_pixels, int mode, const float *b0, const float *b1, float near_t,
                      float far_t, float *rays_o, float *rays_d, float *t, float *cone_scale,
                      ntx_stream stream);
/* The same for ANY set of image-plane locations -- ray_sampler.rays_from_camera (ray_sampler.py:39-48) as Proxy / Frustum call it
 * with whatever the pixel sampler returned (pixel_sampler.py: Full, Independent, Proxy): image_plane_loc DEVICE float32 [n,2] =
 * (row, col) per ray, the tensor the reference casts to float32 (:25, :17).  Outputs as above. */
int ntx_generate_rays_at(const float *c2w, int height, int width, float focal, const float *image_plane_loc, int64_t n_rays, int mode,
                         const float *b0, const float *b1, float near_t, float far_t, float *rays_o, float *rays_d, float *t,
                         float *cone_scale, ntx_stream stream);
/* Replaces proxy.AABB.__call__ (proxy.py:13-35) on the caller's own rays: rays_o, rays_d DEVICE [n,3] -> t DEVICE [n,2],
 * [inf, inf] on a miss; 1/0 and 0*inf follow IEEE as in the reference.  b0, b1: HOST float[3]. */
int ntx_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays, const float *b0, const float *b1, float *t,
                       ntx_stream stream);
/* Same for a STRIDED set of pixels, the local rays of one rank of a shard map (see ntx_shard_count): local ray k is
 * pixel pixel0 + (k / run_length) * run_stride + k % run_length.  rank r of R: pixel0 = r * run_length,
 * run_stride = R * run_length, n_pixels = ntx_shard_count(H * W, run_length, R, r). */
int ntx_generate_rays_strided(const float *c2w, int height, int width, float focal, int64_t pixel0, int64_t n_pixels,
                              int64_t run_length, int64_t run_stride, int mode, const float *b0, const float *b1,
                              float near_t, float far_t, float *rays_o, float *rays_d, float *t, float *cone_scale,
                              ntx_stream stream);

/* Replaces layer.FourierFeatures.call (layer.py:22-23): x[M,D] -> out[M, D*(1+2*n_freq)] (DEVICE). */
int ntx_fourier_features(const float *x, int64_t m, int d, int n_freq, float *out, ntx_stream stream);

/* Replaces model((pos, dirs, params), training) (renderer.py:161; model.py:58-125):
 * pos[M,3], dirs[M,3], params[M,P] -> color[M,3] (raw), sigma[M] (raw alpha head).  All DEVICE. */
int ntx_mlp_forward(ntx_ctx *ctx, const float *pos, const float *dirs, const float *params, int64_t m,
                    uint32_t flags /* NTX_FLAG_FP16X3 or 0 */, float *color_out, float *sigma_out, ntx_stream stream);

/* Replaces Renderer.map_model_output (renderer.py:170-213): color[N,S,3], sigma[N,S], z[N,S],
 * rays_d[N,3] -> color_out[N,3], alpha_out[N], optional weights_out[N,S] (NULL to skip).
 * bkgd: HOST float[3]. */
int ntx_composite(const float *color, const float *sigma, const float *z_vals, const float *rays_d,
                  int64_t n_rays, int n_samples, uint32_t flags, const float *bkgd, float *color_out,
                  float *alpha_out, float *weights_out, ntx_stream stream);

/* Replaces the sample placement of Renderer.render_rays (renderer.py:101-111; MipRenderer :374-383) on its own:
 * z_out[N, n_points] (DEVICE) = t0 (1 - tv) + t1 tv over tv = tf.linspace(0, 1, n_points) evaluated in float32, and with
 * NTX_FLAG_PERTURB the stratified jitter z = lower + (upper - lower) * u of :106-111, u in [0,1) drawn by a counter-based
 * generator: word 0 of Philox4x32-10 at counter (sample index, ray index lo, ray index hi, 0) under the key
 * (perturb_seed lo, hi), low 23 bits as the float32 mantissa (tf.random.uniform's conversion).  The draw depends only on
 * (perturb_seed, ray index, sample index); the ray index is the index within the call unless `opts` maps it to a global one
 * (ntx_render_opts: chunks and shards of one image then draw what the whole image draws under one seed).
 * TensorFlow's own stream (its global generator) cannot be reproduced; the distribution is the same.  ntx_render_rays
 * places its samples with exactly this function. */
int ntx_sample_depths(const float *t, int64_t n_rays, int n_points, uint32_t flags, uint64_t perturb_seed,
                      const ntx_render_opts *opts /* ray index map; may be NULL */, float *z_out, ntx_stream stream);

/* ABI v5.  noise_out[n_rays, n_points] (DEVICE) = raw_noise_std * N(0,1): exactly the draws NTX_FLAG_RAW_NOISE adds to the density inside
 * ntx_render_rays / ntx_render_instanced (ntx_render_opts above: Philox counter (sample, ray index, 1) under `seed`, Box-Muller), as a
 * tensor -- what the reference's `tf.random.normal(alpha.shape) * raw_noise_std` is (renderer.py:190-192).  opts: raw_noise_std and the
 * ray index map.  The training step uses it (config_grass_filtered_train.py:99 trains with raw_noise_std 0.1). */
int ntx_sample_noise(int64_t n_rays, int n_points, uint64_t seed, const ntx_render_opts *opts, float *noise_out, ntx_stream stream);

/* Replaces Renderer.__call__ + render_rays + evaluate_model + map_model_output
 * (renderer.py:47-213) with n_importance=0, fused in one launch:
 * culling of t==inf rays, sample placement, positional encoding, the MLP and the composite.
 *   rays_o[N,3], rays_d[N,3], t[N,2], cone_scale[N] (DEVICE)
 *   params[n_param_rows, P] (DEVICE): ray r uses row r / rays_per_param_row (the reference's
 *       tf.repeat(parameters, HW), renderer.py:54); rays_per_param_row = 1 gives per-ray parameters
 *   blur_idx: -1 = off, else params[blur_idx] *= cone_scale * z per sample (renderer.py:155-158)
 *   flags: NTX_FLAG_MAP_EXR | NTX_FLAG_COMPOSITE_BKGD | NTX_FLAG_CHECK_NUMERICS | NTX_FLAG_FP16X3 | NTX_FLAG_PERTURB | NTX_FLAG_RAW_NOISE
 *   opts: NULL, or HOST ntx_render_opts: raw_noise_std (with NTX_FLAG_RAW_NOISE, keyed by perturb_seed) and the global ray index map
 *   z_vals: NULL, or DEVICE [N,S] sample depths replacing renderer.py:101-111 altogether (wins over NTX_FLAG_PERTURB)
 *   perturb_seed: key of the jitter under NTX_FLAG_PERTURB (renderer.py:106-111, the reference's default; see
 *       ntx_sample_depths), evaluated inside the kernel: no [N,S] tensor exists
 *   status_flag: NULL, or DEVICE int32 OR-ed with 1 when NTX_FLAG_CHECK_NUMERICS finds NaN/Inf
 * Outputs (DEVICE): color_out[N,3] (premultiplied), alpha_out[N]; culled rays get 0 (or bkgd);
 *   weights_out: NULL, or [N,S] the compositing weights of renderer.py:198 (input of sample_pdf; rows of
 *   culled rays are left untouched).
 * All arguments are validated before the first launch: a call that returns an error has written nothing.
 * Context scratch: the compacted list of hit rays, 4 B per ray, sized by ntx_reserve (N beyond it is an error; nothing is
 * allocated here).  The view direction and the appearance parameters are constant along a ray (renderer.py:152-154), so
 * for ParamNerf models the direction segment of the colour layer is evaluated once per ray inside the kernel (through
 * LDS, bit-identical to the per-sample evaluation) unless blur_idx scales an appearance parameter.  Calls on one context
 * must be stream-ordered. */
int ntx_render_rays(ntx_ctx *ctx, const float *rays_o, const float *rays_d, const float *t,
                    const float *params, int64_t rays_per_param_row, const float *cone_scale,
                    int64_t n_rays, int n_samples, int blur_idx, uint32_t flags, const float *bkgd,
                    const float *z_vals, uint64_t perturb_seed, const ntx_render_opts *opts, float *color_out,
                    float *alpha_out, float *weights_out, int32_t *status_flag, ntx_stream stream);

/* Replaces the importance-sampling step of Renderer.render_rays (renderer.py:125-130) incl. sample_pdf
 * (renderer.py:589-617): bins = midpoints of the coarse depths, pdf = weights[:,1:-1] + 1e-5, n_importance
 * depths by inverse CDF at u (NULL = tf.linspace(0,1,n_importance), the `det` branch; else DEVICE [N,n_imp]
 * uniform draws), merged with the coarse depths and sorted -> z_out[N, S + n_importance] (DEVICE).
 * The coarse depths are z_vals[N,S], or (NULL) recomputed from t exactly as ntx_render_rays places them under the same
 * `flags` (NTX_FLAG_PERTURB or 0), `perturb_seed` and ray index map. */
int ntx_sample_pdf(const float *t, const float *z_vals, const float *weights, const float *u, int64_t n_rays,
                   int n_samples, int n_importance, uint32_t flags, uint64_t perturb_seed,
                   const ntx_render_opts *opts /* ray index map; may be NULL */, float *z_out, ntx_stream stream);

/* Replaces InstanceRenderer.evaluate_model + map_model_output (renderer.py:247-354) DOWNSTREAM of the
 * instancer: the arguments are the buffers instancer.get_model_input returns (instancer.pyx:38-54), on the
 * device:  rays_d_map[N,S,3], pts[N,S,3], t[N,S], dists[N,S], color_last[N,3], alpha_last[N],
 * alpha_weight[N,S] (NULL = density_reweighting off), instance_id[N,S] int32, hit[N] uint8 (the rays in
 * `idxs`), params_map[N,S,P], cone_scale[N].  Samples with dists <= 0 are skipped (renderer.py:284-288);
 * sigma *= alpha_weight * density_scale (:300); alpha = 1 - exp(-relu(sigma) * dists / patch_scale) (:339);
 * one extra sample (color_last as is, alpha_last as an alpha) closes every ray (:331,339).
 * instance_color[n_instances,3] != NULL = the false-colour mode (:306-307).  Rays with hit == 0 get 0, also
 * under NTX_FLAG_COMPOSITE_BKGD (:313-314).  1 <= n_samples <= 4096.  flags: as ntx_render_rays without NTX_FLAG_PERTURB
 * (the instancer places the samples).  opts: NULL, or raw_noise_std + noise_seed (NTX_FLAG_RAW_NOISE, added to the scaled
 * density of the in-patch samples: at dists == 0 the reference's draw has no effect either) and the ray index map. */
int ntx_render_instanced(ntx_ctx *ctx, const float *rays_d_map, const float *pts, const float *t,
                         const float *dists, const float *color_last, const float *alpha_last,
                         const float *alpha_weight, const int32_t *instance_id, const uint8_t *hit,
                         const float *params_map, const float *cone_scale, int64_t n_rays, int n_samples,
                         int blur_idx, float patch_scale, float density_scale, uint32_t flags,
                         const float *bkgd, const float *instance_color, const ntx_render_opts *opts, float *color_out,
                         float *alpha_out, int32_t *status_flag, ntx_stream stream);

/* Replaces the image post-processing of logger.Logger.render_image / write_image (logger.py:128-144) and
 * util.interpolate.filtered_downsample (interpolate.py:68-82): rgba[H,W,4] premultiplied (DEVICE) ->
 * optional gaussian low-pass (size 3*factor, std factor/2) + stride-`factor` downsample with TF 'SAME' padding,
 * optional rgb / (a + 1e-5), written as float32 out_f32[ceil(H/f),ceil(W/f),4] and/or uint8 out_u8 (saturating
 * x*255.5, what tf.image.convert_image_dtype does before encode_png).  Either output may be NULL. */
int ntx_image_epilogue(const float *rgba, int height, int width, int downsampling_factor, int unpremultiply,
                       float *out_f32, uint8_t *out_u8, ntx_stream stream);

/* ---- multi-GPU: rays shard embarrassingly, the ONE collective is the gather of the finished RGBA --------------
 * The reference has no distribution; Renderer.__call__ already renders chunks of rays independently (renderer.py:72-73).
 * Shard map (ntx_shard_*): the row-major pixel sequence [0, n_pixels) (pixel_sampler.py:14-15) is cut into RUNS of
 * `run_length` consecutive pixels (the last one may be shorter); run q belongs to rank q % n_ranks and is that rank's
 * local run q / n_ranks.  run_length = ceil(n_pixels / n_ranks) gives contiguous bands; run_length = image width deals
 * rows round-robin, which balances the rays the proxy culls (renderer.py:58-67) across ranks. */
int64_t ntx_shard_count(int64_t n_pixels, int64_t run_length, int n_ranks, int rank);   /* pixels of `rank`, -1 on bad arguments */

/* One communicator per process/device over RCCL (librccl is dlopen-ed on first use: the copy PyTorch already loaded if
 * there is one).  ntx_comm_unique_id wraps ncclGetUniqueId (rank 0 calls it and hands the 128 bytes to its peers by any
 * host-side means); ntx_comm_create wraps ncclCommInitRank on `device`; all ranks call it collectively. */
typedef struct ntx_comm ntx_comm;
#define NTX_COMM_ID_BYTES 128
/* ABI v3.  ntx_comm_preflight: everything of ntx_comm_create that can fail WITHOUT a peer (librccl loadable with every symbol,
 * `device` valid), so that ranks can agree to go ahead before any of them blocks in ncclCommInitRank.  ntx_comm_library: path of the
 * librccl the symbols were bound from ("" = none). */
int ntx_comm_preflight(int device);
const char *ntx_comm_library(void);
int ntx_comm_unique_id(uint8_t *id_out /* HOST [NTX_COMM_ID_BYTES] */);
int ntx_comm_create(const uint8_t *id /* HOST [NTX_COMM_ID_BYTES] */, int n_ranks, int rank, int device, ntx_comm **out);
int ntx_comm_destroy(ntx_comm *comm);

/* Replaces nothing in the reference (single device); it is the `tf.concat` of the chunk results (renderer.py:76-79)
 * across devices.  Every rank passes its local premultiplied RGBA, local_rgba[ntx_shard_count(...), 4] (DEVICE), in local
 * ray order; `root` receives the whole image image_out[n_pixels, 4] (DEVICE) in pixel order.  One ncclGather
 * (rccl.h:745) on `stream` when every rank holds the same number of pixels -- each peer sends straight to the root over its own xGMI
 * link -- else the same exchange as grouped ncclSend/ncclRecv with the exact counts.  staging (root only, DEVICE, may be
 * NULL when run_length == ceil(n_pixels / n_ranks) and the counts are equal: the gather then lands in image_out
 * directly): n_ranks * max-count * 4 floats, from which a kernel deals the runs back into pixel order.
 * image_out / staging are ignored on the other ranks. */
int ntx_gather_image(ntx_comm *comm, const float *local_rgba, int64_t n_pixels, int64_t run_length, float *image_out,
                     float *staging, int root, ntx_stream stream);
/* ABI v7.  The same with flags.  NTX_GATHER_FORCE_EXCHANGE: take the exact-count branch whatever the counts are -- grouped
 * ncclSend / ncclRecv into `staging` (required on the root), then the un-shard pass -- and send the root's own block to itself too
 * instead of copying it: a communicator of ONE rank then executes every RCCL call of the branch that uneven shards take on eight
 * (tests; a box with one GPU).  ntx_comm_version: ncclGetVersion of the bound librccl (0 = none). */
#define NTX_GATHER_FORCE_EXCHANGE 1u
int ntx_gather_image_ex(ntx_comm *comm, const float *local_rgba, int64_t n_pixels, int64_t run_length, float *image_out,
                        float *staging, int root, uint32_t flags, ntx_stream stream);
int ntx_comm_version(void);

/* ABI v5.  values[n] (DEVICE) <- the mean over all ranks of their values[n], in place, on `stream`: one ncclAllReduce (rccl.h) and a scale.
 * The one collective of data-parallel TRAINING (the reference trains on one device: train.py:61-67): ranks take the same step on different
 * rays, the loss is a mean over rays, so the full batch's gradient is the mean of the shards' (equal shard sizes).  ntx_comm_size: n_ranks. */
int ntx_comm_size(const ntx_comm *comm);
int ntx_allreduce_mean_f32(ntx_comm *comm, float *values, size_t n, ntx_stream stream);

/* ABI v3, host only (no device, no communicator): the exchange ntx_gather_image performs for a shard map, as numbers -- rank r
 * holds counts_out[r] pixels and its block starts at pixel slot offsets_out[r] of the destination; *equal_out = one ncclGather
 * (else grouped Send/Recv with the exact counts); *direct_out = the destination is image_out itself (else `staging`, followed by
 * the un-shard pass).  ntx_unshard_map: src_out[p] = the staging pixel slot that holds pixel p (what the un-shard kernel reads).
 * Both run the very code of the device path (csrc/ntx_shard.h), so CPU tests can execute the plan through another transport. */
int ntx_gather_plan(int64_t n_pixels, int64_t run_length, int n_ranks, int64_t *counts_out /* [n_ranks] or NULL */,
                    int64_t *offsets_out /* [n_ranks] or NULL */, int *equal_out, int *direct_out);
int ntx_unshard_map(int64_t n_pixels, int64_t run_length, int n_ranks, int64_t *src_out /* HOST [n_pixels] */);

/* Introspection for benches/tests: name and launch geometry of the fused kernel in `ctx`. */
int ntx_kernel_info(ntx_ctx *ctx, int *n_workgroups, int *threads_per_workgroup, int *n_cus);

/* Host-only helpers (no device needed) exposing the weight packing, so it can be checked on CPU:
 * number of floats of the packed image, and the packing itself. */
size_t ntx_packed_count(const ntx_model_desc *desc);
int ntx_pack_weights(const ntx_model_desc *desc, const float *weights_host, size_t n_floats,
                     float *packed_out, size_t n_packed);

/* Same for the fp16x3 weight stream: its size in bytes (0 + error for families without one) and the packing. */
size_t ntx_packed_fp16x3_bytes(const ntx_model_desc *desc);
int ntx_pack_weights_fp16x3(const ntx_model_desc *desc, const float *weights_host, size_t n_floats,
                            uint16_t *packed_out, size_t n_bytes);

/* ---- ABI v4: the patch instancer (what feeds ntx_render_instanced) ---------------------------------------------------------
 * Replaces instancer.instancer.Instancer (instancer/instancer.pyx:6-54) = C_Instancer (instancer/src/instancer.hpp:10-89), the
 * reference's only native code: Embree 3 on one CPU thread.  Built here: the constructor with explicit `transformations`
 * (instancer.pyx:19-20 -> AddInstance, instancer.cpp:124-141), an instancer mesh given as arrays, GetNumberOfInstances (:426-428),
 * the matrices ExportTransformations writes (:1040-1061) and GetModelInput (:751-1037) with the three patch choices, mean
 * distances, directional and point lights, shadow rays (:591-602, 945-961, 1018-1027), auxiliary meshes with
 * their flat shading (:393-417, 716-743); since ABI v5 image textures, as parameters on the instancer mesh (:640-667) and as the
 * albedo of an auxiliary mesh (:725-733), see below.  DistributeInstancesOnMesh (:233-390) is setup on the host: its result is the
 * transformation list ntx_instancer_create takes (nerf_tex_amd.instancer.distribute_instances_on_mesh restates it; the reference
 * can also export its own with `transformation_export_path`).
 *
 * ntx_instancer_desc = the constructor arguments that survive (instancer.cpp:53-93): b_0 / b_1 the patch box in patch
 * coordinates; n_parameters, light_dir_parameter_idx, light_strength_parameter_idx as the `textures` list defines them
 * ("" = 1 parameter, "light" = 3 with light_dir at their start, "point" = 4: strength, then position = light_dir + 1;
 * -1 = none); instance_sample_method 0 random / 1 nearest / 2 nearest_blend (instancer.pyx:14); patch_scale only scales
 * nearest_blend's transition range (:697; 1 unless the patches were distributed on a mesh).  cast_shadow_rays (needs a light
 * entry): a sample whose shadow query is occluded gets the light direction (0, 0, -1) (:571-573).  A query (isShadowed, :591-602:
 * from the point along the light parameter AS GIVEN -- for 'point' that is the light's position, :956 -- with 0 < t <= 100) is
 * occluded by the top face of a patch box entered from outside, by its bottom face either way, by a mesh hit from its front and
 * by the triangle with primID 1 of any mesh from either side (filter :543-554: its `primID == 1` clause does not ask which geometry).  With N = max(min_shadow_samples, n_shadow_samples * total length) < n_pts the queries are
 * made at max(min_shadow_samples, N * length / total) points spaced evenly along every segment and a step takes the nearer of
 * the two around it (:946-958, 1018-1027), else every step makes its own (:959-961).  min_shadow_samples >= 2.
 * *status_flag |= 4 when a ray needed more than 4096 shadow samples (the rest read as unshadowed). */
typedef struct ntx_instancer ntx_instancer;
typedef struct ntx_instancer_desc {
    uint32_t size;                       /* sizeof(ntx_instancer_desc) */
    float b_0[3], b_1[3];
    int32_t n_parameters, light_dir_parameter_idx, light_strength_parameter_idx;
    int32_t instance_sample_method, use_mean_distance, cast_shadow_rays;
    float patch_scale;
    int32_t min_shadow_samples, n_shadow_samples;   /* with cast_shadow_rays (instancer.cpp:53, 861, 1019) */
} ntx_instancer_desc;
#define NTX_INSTANCER_DEFAULT_MAX_RAYS (1 << 16)
/* transformations: HOST [n_instances,4,4] row-major patch -> world, what AddInstance takes.  The instancer keeps, per instance,
 * the inverse (world -> patch), the direction map (columns of the 3x3 block, normalised) and the origin (instancer.cpp:127-132;
 * computed in double and rounded), plus hit-list workspace for NTX_INSTANCER_DEFAULT_MAX_RAYS rays (3.2 KB per ray). */
int ntx_instancer_create(const ntx_instancer_desc *desc, const float *transformations, int64_t n_instances, int device,
                         ntx_instancer **out);
int ntx_instancer_destroy(ntx_instancer *inst);
/* Workspace for calls of up to max_rays rays in one piece (<= 2^24); larger calls are cut into pieces of the reserved size. */
int ntx_instancer_reserve(ntx_instancer *inst, int64_t max_rays);
int64_t ntx_instancer_count(const ntx_instancer *inst);                       /* GetNumberOfInstances; -1 on NULL */
/* HOST outputs, each may be NULL: world_to_patch[K,4,4] (this->transformations; its inverses are what ExportTransformations
 * writes), directions[K,3,3] (dir_transformations), origins[K,3] (instance_origins). */
int ntx_instancer_matrices(const ntx_instancer *inst, float *world_to_patch, float *directions, float *origins);
/* The instancer mesh (instancer.cpp:369-389: in the scene for culling): HOST vertices[n_vertices,3], faces[n_faces,3].  A ray
 * ends at its closest crossing of the mesh and is closed by an opaque black sample (:1013-1016).  n_faces = 0 removes it. */
int ntx_instancer_set_mesh(ntx_instancer *inst, const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces);
/* The same with auxiliary meshes (AddMesh, instancer.cpp:393-417) in one list: face_kind[f] bit 0 = 0 for the instancer mesh (black closing
 * sample), 1 for an auxiliary mesh; bit 1 (value 2) = the face is primID 1 of its own mesh (the shadow filter's clause above; without
 * face_kind the list is one mesh and face 1 has it).  An auxiliary mesh's closing sample is shaded (shadeMesh, :716-743): albedo 0.8 * min(diffuse + 0.2, 1), diffuse =
 * max(n . l, 0) with the interpolated vertex normal n (normals[n_vertices,3], HOST) and the light parameter l, 0 when the point just
 * above the surface is shadowed (isShadowed).  Needs a light entry in the textures list; textures: ntx_instancer_set_mesh_textures.  Every mesh
 * culls and casts shadows alike.  normals / face_kind may be NULL (= ntx_instancer_set_mesh). */
int ntx_instancer_set_meshes(ntx_instancer *inst, const float *vertices, const float *normals, int64_t n_vertices, const int32_t *faces,
                             const uint8_t *face_kind, int64_t n_faces);
/* GetModelInput (instancer.cpp:751-1037) with the buffers of get_model_input (instancer.pyx:38-54), all DEVICE, every element of
 * every output written: rays_o[N,3], rays_d[N,3] (per ray; the reference's [N,S,3] input is this row repeated), parameters[N,P] ->
 * rays_d_map[N,S,3], pts[N,S,3], t[N,S], dists[N,S], color_last[N,3], alpha_last[N], alpha_weight[N,S] (density_weight),
 * instance_id[N,S], hit[N] uint8, params_map[N,S,P]: the argument list of ntx_render_instanced.  Per ray: face crossings of every
 * instanced box with 0 < t <= 100 (at most 200 kept, instancer.cpp:22), sorted by (t, instance); the union of the boxes is
 * marched in steps of step_size from a random offset; a sample takes the patch it lies in (several: by
 * instance_sample_method) and is mapped into it (point, direction, light direction / strength); samples behind the last
 * step keep the defaults of instancer.pyx:41-50.  The reference draws from one std::mt19937 in ray order (:855, 675, 710);
 * here the offset of a ray is U[0,1) from word 0 of Philox4x32-10 at counter (0, ray lo, ray hi, 2) under `seed`, the patch
 * choice of (ray, step) from counter (step, ray lo, ray hi, 3); `opts` carries the ray index map (as for ntx_render_rays;
 * NULL = the call's own indices), so chunked or sharded calls draw what the whole image draws.
 * An instancer belongs to one device and owns one workspace: calls on it must be stream-ordered (like a context's).
 * *status_flag (DEVICE, may be NULL) |= 1 when a ray had more than 200 face crossings (the rest were dropped, which ones is
 * unspecified -- as in the reference).  1 <= n_pts <= 4096. */
int ntx_instancer_model_input(ntx_instancer *inst, const float *rays_o, const float *rays_d, const float *parameters, int64_t n_rays,
                              int n_pts, float step_size, uint64_t seed, const ntx_render_opts *opts, float *rays_d_map, float *pts,
                              float *t, float *dists, float *color_last, float *alpha_last, float *alpha_weight,
                              int32_t *instance_id, uint8_t *hit, float *params_map, int32_t *status_flag, ntx_stream stream);

/* ---- ABI v5: image textures of the instancer --------------------------------------------------------------------------------
 * ntx_texture = ONE channel matrix as loadTexture builds it (instancer.cpp:34-50): value / 255, element (r, c) at texels[r * cols + c]
 * with r = the pixel's column x (rows = image width) and c = its row counted from the BOTTOM of the image (cols = image height);
 * HOST memory, copied.  Lookups are interpolate2d (:605-625): bilinear around x * (rows - 1, cols - 1), indices by truncation,
 * weights x - floor(x); an index outside the matrix is clamped (the reference reads past it; at u or v = 1 that weight is 0). */
typedef struct ntx_texture {
    const float *texels;
    int32_t rows, cols;
} ntx_texture;
/* Parameter textures (getParameters, instancer.cpp:640-667; the constructor's image entries, :84-88, after DistributeInstancesOnMesh
 * -- without a mesh_path the reference loads and counts them and never applies them, :911: then do not call this).  The instancer
 * mesh as the reference holds it: HOST vertices[n_vertices,3], uv[n_vertices,2], faces[n_faces,3].  Texture file i multiplies
 * parameter parameter_idx[i] by textures[i] at the texture coordinates of the closest point of the mesh (closest_point_triangle,
 * :154-198) strictly within patch_max_extent (:69 scaled by :246) of the sample; no triangle that close: unchanged.  The caller
 * resolves the reference's indexing: its list holds every CHANNEL of every file and file i uses entry i of that list (:656-662),
 * so a file of c channels takes c parameters and has one of them multiplied.  With N = max(min_texture_samples,
 * n_texture_samples * total length) < n_pts a ray looks the textures up at max(min_texture_samples, N * length / total) points
 * spaced evenly along every segment and a step's WHOLE parameter row is s0 * (1 - w) + s1 * w between the two around it (:910-923,
 * 989-998; the light entries are written afterwards); else every step makes its own lookup (:926).  At most 4 files;
 * 2 <= min_texture_samples <= 512.  n_textures = 0 removes them.  Of several triangles at one distance the lowest index wins
 * (Embree's order is its BVH's). */
int ntx_instancer_set_parameter_textures(ntx_instancer *inst, const float *vertices, const float *uv, int64_t n_vertices, const int32_t *faces,
                                         int64_t n_faces, float patch_max_extent, int n_textures, const int32_t *parameter_idx,
                                         const ntx_texture *textures, int min_texture_samples, int n_texture_samples);
/* Albedo textures of auxiliary meshes (AddMesh's texture_path, instancer.cpp:404; shadeMesh :725-733), after
 * ntx_instancer_set_meshes and for the same list: uv[n_vertices,2]; face_texture[n_faces] = the texture set of the face's mesh
 * or -1 (albedo 0.8); textures[3 * n_sets]: three channel matrices per set (an image that does not have exactly three channels
 * gives its first for all three, :732).  n_sets = 0 removes them. */
int ntx_instancer_set_mesh_textures(ntx_instancer *inst, const float *uv, int64_t n_vertices, const int32_t *face_texture, int64_t n_faces,
                                    int n_sets, const ntx_texture *textures);

/* ---- ABI v5: one training step (f6) ------------------------------------------------------------------------------------------
 * Replaces the body of the reference's training loop, network/train.py:61-67: `pred = renderer(**data)` under a GradientTape, a loss
 * of network/loss.py:6-59, `tape.gradient`, `optimizer.apply_gradients` with tf.keras.optimizers.Adam under ExponentialDecay
 * (train.py:49-52).  Built for the architecture of the shipped training configs: ParamNerf, depth 8, width 256, skips [4],
 * color_depth 1, Fourier features (any n_parameters, any band counts); NTX_E_UNSUPPORTED otherwise.  The trainer owns the weights
 * (Keras get_weights() order, like ntx_create), Adam's moments, the gradient and every layer's activations for up to
 * max_rays x max_samples_per_ray samples (<= 1024 samples per ray; 23 KB per sample: 6 GB for the configs' 4 x 256 x 256; pos_map and
 * dir_map at most 96 features wide each).  Everything float32.  A step is bit-reproducible: weight gradients are summed over the samples
 * in a fixed order.  Against float64 autograd of the restated step (oracle/train_oracle.py, following the float32 pass's ReLU branches) every
 * layer's gradient, the loss and the predictions are within 1e-4 rel-Linf at the configs' batch; where a batch is ill-conditioned (a handful
 * of coarse steps per ray, depths a hair apart) within 4 times what float32 autograd of the same restatement is off by. */
typedef struct ntx_trainer ntx_trainer;
#define NTX_LOSS_NERF 0                  /* network.loss.NerfLoss  (loss.py:6-19):  loss_fn(color_true, color_pred) */
#define NTX_LOSS_ALPHA 1                 /* network.loss.AlphaLoss (loss.py:21-49): + gamma * alpha_loss_fn(alpha_true, alpha_pred), colours masked by alpha_true */
#define NTX_LOSS_MSE 0                   /* network.loss.mse   (loss.py:51-54) */
#define NTX_LOSS_SMAPE 1                 /* network.loss.smape (loss.py:56-59), eps = 1e-2 */
typedef struct ntx_loss_desc {
    uint32_t size;                       /* sizeof(ntx_loss_desc) */
    int32_t kind, loss_fn, alpha_loss_fn;
    float gamma;                         /* AlphaLoss: 1 */
    int32_t filter_color_loss, use_hard_mask;   /* AlphaLoss: True, True */
} ntx_loss_desc;
int ntx_trainer_create(const ntx_model_desc *desc, const float *weights_host, size_t n_floats, int device, int64_t max_rays, int max_samples_per_ray,
                       ntx_trainer **out);
int ntx_trainer_destroy(ntx_trainer *t);
size_t ntx_trainer_weight_count(const ntx_trainer *t);
#define NTX_TRAINER_WEIGHTS 0
#define NTX_TRAINER_GRADIENTS 1
#define NTX_TRAINER_ADAM_M 2
#define NTX_TRAINER_ADAM_V 3
/* Copies one of the trainer's weight-shaped vectors to HOST memory (Keras get_weights() order; synchronises the device). */
int ntx_trainer_get(ntx_trainer *t, int what, float *out_host, size_t n_floats);
int ntx_trainer_set_weights(ntx_trainer *t, const float *weights_host, size_t n_floats);
/* ... any of the four, from HOST memory (NTX_TRAINER_GRADIENTS: a gradient averaged over ranks by another transport than RCCL). */
int ntx_trainer_set(ntx_trainer *t, int what, const float *values_host, size_t n_floats);
/* Data-parallel training: every rank's gradient <- the mean over the ranks (ntx_allreduce_mean_f32 on the trainer's own buffer), between
 * ntx_train_step_gradients and ntx_trainer_adam_step; all ranks then take the same Adam step. */
int ntx_trainer_allreduce_gradients(ntx_trainer *t, ntx_comm *comm, ntx_stream stream);
/* The activations the last step kept, to HOST memory as [n_samples_total][width]: layer 0-7 = the trunk layers' outputs (after their ReLU),
 * 8 / 9 = the two colour layers' (width 256 / 128), 10 = the raw density (width 1), 11 = the raw colour (width 3).  Tests hand their signs to the float64 restatement, so
 * that its autograd follows the ReLU branches the float32 forward took (a pre-activation within rounding of zero can fall either way).
 * 20-27 = the GRADIENTS the last step kept at the trunk layers' outputs (behind their ReLU: what the layer's weight gradient contracts
 * with), 28 = at the first colour layer's output, 29 = at the feature layer's (all width 256): tests compare them row by row.
 * 30 = the composite's adjoint (width 4): dL/d raw colour, dL/d raw density per sample, where the way back starts. */
int ntx_trainer_activation(ntx_trainer *t, int layer, int64_t n_samples_total, float *out_host);
/* Forward (Renderer.__call__, renderer.py:47-213 -- a ray whose tnear_far is inf, i.e. that misses the proxy, stays in the batch and predicts
 * 0 / the background with alpha 0 as the reference's filter-and-scatter makes it, :58-86 -- : sample depths by ntx_sample_depths -- NTX_FLAG_PERTURB /
 * perturb_seed / opts as there -- or given as z_vals[N,S]; encodings; the network; map_model_output with NTX_FLAG_MAP_EXR /
 * NTX_FLAG_COMPOSITE_BKGD), the loss, and its gradient with respect to every weight, left in the trainer (ntx_trainer_get /
 * ntx_trainer_adam_step).  DEVICE: rays_o[N,3], rays_d[N,3], tnear_far[N,2] (or NULL with z_vals), params[rows,P] (ray r uses row
 * r / rays_per_param_row), cone_scale[N] (blur_idx >= 0), color_true[N,3], alpha_true[N] (AlphaLoss); outputs, each may be NULL:
 * color_pred[N,3], alpha_pred[N], loss_out[1].  bkgd: HOST [3] or NULL (white). */
int ntx_train_step_gradients(ntx_trainer *t, const float *rays_o, const float *rays_d, const float *tnear_far, const float *params, int64_t rays_per_param_row,
                             const float *cone_scale, int64_t n_rays, int n_samples, int blur_idx, uint32_t flags, const float *bkgd, uint64_t perturb_seed,
                             const ntx_render_opts *opts, const float *z_vals, const float *color_true, const float *alpha_true, const ntx_loss_desc *loss,
                             float *color_pred, float *alpha_pred, float *loss_out, ntx_stream stream);
/* optimizer.apply_gradients (train.py:67): Adam (m += (g - m)(1 - beta_1); v += (g^2 - v)(1 - beta_2); w -= lr_t m / (sqrt(v) + epsilon),
 * lr_t = lr sqrt(1 - beta_2^t) / (1 - beta_1^t), t = iterations + 1: TF 2.4's ApplyAdam) with lr = lrate * lrate_decay_rate ^
 * (iterations / lrate_decay_steps) when lrate_decay_steps > 0 (ExponentialDecay, train.py:49-50: decay_steps = lrate_decay * 1e3,
 * decay_rate 0.1), else lrate.  Keras defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7.  Counts the iteration. */
int ntx_trainer_adam_step(ntx_trainer *t, float lrate, float lrate_decay_steps, float lrate_decay_rate, float beta_1, float beta_2, float epsilon,
                          ntx_stream stream);
int64_t ntx_trainer_iterations(const ntx_trainer *t);
/* ABI v6.  Resuming a run (train.py:55-60, logger.py:30-39: the reference checkpoints model + step + optimizer and continues with
 * `train_dataset.take(n_iters - logger.step)`): Adam's iteration count -- what its bias correction and the ExponentialDecay schedule run
 * on -- is set beside the weights and moments (ntx_trainer_set). */
int ntx_trainer_set_iterations(ntx_trainer *t, int64_t iterations);
/* ABI v6.  A coarse and a fine pass (renderer.py:125-138, loss.py:15-16, 41-47, model.py:47-56): the coarse step also leaves the composite's
 * weights a_i T_i -- what sample_pdf takes, with no gradient through it (:129) -- in weights_dev[N,S] (DEVICE; NULL: no more); ntx_sample_pdf
 * places the fine depths, a step with z_vals takes them.  When both passes run on ONE network (model_fine is None, :132) its gradient is the
 * sum of the two steps': op 0 keeps the gradient of the step just taken, op 1 adds it to the next one's. */
int ntx_trainer_composite_weights(ntx_trainer *t, float *weights_dev);
int ntx_trainer_stash_gradients(ntx_trainer *t, int op, ntx_stream stream);
/* ABI v6.  ntx_set_weights from DEVICE memory (the context's device; Keras get_weights() order), in the stream's order and without a
 * host copy: the float32 weight image is remade by one gather kernel.  The validation render inside the reference's training loop
 * (logger.py:76-81 from train.py:61-70) is this with the trainer's weights (ntx_trainer_device_weights).  The fp16x3 images are NOT
 * remade: NTX_FLAG_FP16X3 is refused (NTX_E_UNSUPPORTED) until the next ntx_set_weights. */
int ntx_set_weights_device(ntx_ctx *ctx, const float *weights_dev, size_t n_floats, ntx_stream stream);
/* The trainer's weights where they live: DEVICE memory of the trainer's device, Keras get_weights() order, ntx_trainer_weight_count floats,
 * valid until ntx_trainer_destroy; steps on a stream change them in that stream's order.  What ntx_set_weights_device takes. */
int ntx_trainer_device_weights(ntx_trainer *t, const float **weights_dev);
/* A dense contraction (f32 MFMA, 128 x 128 x 16 tiles through LDS) on DEVICE buffers, for tests and benches (round 4's trainer was made of it;
 * the step now runs on the chain and weight-gradient kernels of csrc/ntx_train_device.h):
 * C[M][N] = op(A) . op(B) (+ bias[N]) (ReLU); a_kcontig: A is [M][K] (row stride lda), else [K][M]; B is [K][N] (b_kcontig must be 0). */
int ntx_gemm_f32(const float *A, int lda, int a_kcontig, const float *B, int ldb, int b_kcontig, float *C, int ldc, int M, int N, int K, const float *bias,
                 int relu, ntx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFTEX_H */
