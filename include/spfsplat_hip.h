/*
 * spfsplat_hip.h -- C ABI of the MI355X (gfx950) Gaussian-splat rasterizer + RoPE-2D library
 * (libspfsplat_hip.so).
 *
 * Plain pointers and sizes only: no torch / pybind types.  Every pointer is a DEVICE pointer
 * unless its comment says "host".  The library never allocates device memory, never
 * synchronises the device (except spf_stage_times_ms, which the caller asks for) and launches
 * everything on the stream it is given, so it can be driven from any host language.
 *
 * What each entry point replaces in the reference (ranrhuang/SPFSplatV2):
 *
 *   spf_raster_*            the external rasterizer behind
 *                           `GaussianRasterizer(settings)(means3D=..., viewmatrix=...)`
 *                           src/model/decoder/cuda_splatting.py:105-138 (forward) and its autograd
 *                           backward; package diff_gauss_pose, requirements.txt:88.  One call here
 *                           covers a whole batch of (scene, view) renders, i.e. the Python loop at
 *                           cuda_splatting.py:96-143 and the per-view `repeat` copies at
 *                           src/model/decoder/decoder_splatting_cuda.py:59-64.
 *   spf_camera_*            the camera preparation inside render_cuda: scale-invariant rescale
 *                           (cuda_splatting.py:66-74), get_fov (src/geometry/projection.py:269-283),
 *                           get_projection_matrix (cuda_splatting.py:15-42), extrinsics.inverse() and the
 *                           transposes (cuda_splatting.py:84-91), and its autograd backward to the poses.
 *   spf_adapter_*           UnifiedGaussianAdapter.forward (src/model/encoder/common/gaussian_adapter.py:122-150)
 *   spf_rope2d              `rope_2d(tokens, positions, base, fwd)` (and VGGT's RotaryPositionEmbedding2D,
 *                           src/model/encoder/backbone/vggt/layers/rope.py:62-188, same rotation out of place)
 *                           src/model/encoder/backbone/croco/curope/curope.cpp:49-65 and
 *                           curope/kernels.cu:84-108 (in place, forward and backward).
 *
 * Return value of every int function: 0 = success, otherwise a negative SPF_E_* code;
 * spf_last_error() returns a host string describing the most recent failure on this thread.
 */
#ifndef SPFSPLAT_HIP_H
#define SPFSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPF_ABI_VERSION 6

#define SPF_OK 0
#define SPF_E_INVALID (-1)   /* bad argument (null pointer, size, unsupported degree ...) */
#define SPF_E_LAUNCH (-2)    /* a HIP call failed; see spf_last_error() */
#define SPF_E_CAPACITY (-3)  /* pair buffer smaller than the number of (Gaussian, tile) pairs */

#define SPF_UNKNOWN 0xffffffffu
#define SPF_TILE 16          /* square tile edge in pixels (four waves per tile); spf_raster_num_tiles() counts them */
#define SPF_DENSE_AREA 26    /* mean cull-box area (px) above which a tile's BACKWARD takes the dense "rows" form inside the
                                compositing kernel (and what the dense-tile census, counters[3], counts) */
#define SPF_DENSE_AREA_FWD 120 /* the same for the FORWARD: the sparse "lists" form stays ahead of the rows form up to much
                                denser tiles than in the backward (round 5 sweep), so a tile may composite forward through
                                lists and backward through rows -- both write / read the same per-pixel state */

/* Geometry of one batched call: S scenes, V views each => R = S*V renders of H x W pixels.
 * All scenes hold G Gaussians with K SH coefficients per colour channel (stride); the SH basis is
 * evaluated up to min(sh_degree, 3), or min(sh_degree, 4) when sh_band4 is set.  K == 0 means colours are given
 * directly (colors_precomp). */
typedef struct SpfDims {
    int32_t S, V, G, K, sh_degree, H, W;
    float scale_modifier;
    int32_t sh_layout;   /* 0: shs / dL_dshs are [S,G,K,3] (what the reference hands its rasterizer,
                            cuda_splatting.py:79); 1: [S,G,3,K] (the encoder's native layout: no transposed copy);
                            2 ("band split", K = 25 only): TWO planes, shs / dL_dshs = [S,G,3,16] (bands 0 - 3) and
                            shs_high / dL_dshs_high = [S,G,3,9] (band 4) -- what spf_adapter_forward writes when it is given
                            a second plane.  The reference ships d_sh = 25 (config/model/encoder/spfsplatv2.yaml:20) and
                            a degree-3 evaluation (sh_band4 = 0) then touches plane 0 only: the band-4 third of every
                            coefficient block neither crosses HBM in the forward nor is zero-written in the backward
                            (in one [3,25] block it shares cache lines with the bands that are read);
                            3 ("raw rows"): the ADAPTER IS FUSED INTO THE PROJECTION KERNELS -- scales, rotations and shs are
                            NULL and every Gaussian is one row of SpfInputs.raw, the network's 7 + 3K raw channels
                            (UnifiedGaussianAdapter.forward, gaussian_adapter.py:122-150): the kernels form
                            scales = min(0.001 softplus(raw[0:3]), 0.3), rotations = raw[3:7] / (|.| + adapter_eps) and
                            sh[c][k] = raw[7 + c K + k] * sh_mask[k] as they read the row, and the backward chains through
                            them into dL_draw -- the adapter's own pass over 656 + 576 bytes per Gaussian and step (more
                            than the decoder's) never happens.  The same expressions in the same order as
                            spf_adapter_forward -> sh_layout 1: results agree to float32 rounding */
    int32_t sh_band4;    /* 0 (default): the reference's d_sh = 25 / sh_degree 4 (config/model/encoder/spfsplatv2.yaml:20,
                            cuda_splatting.py:77-78,114) is accepted as a stride and evaluated to degree 3 like the
                            published 3DGS kernels; 1: band 4 (coefficients 16..24) is evaluated too, forward and
                            backward (the `pose` fork's behaviour is not knowable offline: SURVEY.md 0.6) */
    int32_t bin_cap;     /* 0: tile lists are packed (tile_start = exclusive scan of the tile counts; the classic chain
                            project -> scan -> bin -> sort).  > 0 ("direct bins", planned calls only): every tile owns a
                            fixed bin of bin_cap entries at pairs[tile * bin_cap ..] and the PROJECTION kernel itself
                            writes the keys there -- no scan, no binning pass, two launches and one pass over the
                            (Gaussian, view) pairs fewer.  The caller sizes st->pairs for S*V*tiles*bin_cap entries
                            (<= 2^31) and passes the same value to every call of the forward/backward pair; a tile that
                            needs more than bin_cap entries raises plan flag 2 (see spf_raster_forward_render). */
    int64_t pair_capacity; /* direct bins only: number of gradient records g->gpair will hold (the `capacity` the
                            backward is given); the projection kernel numbers the (Gaussian, tile) pairs and raises plan
                            flag 1 if they do not fit -- per SHARD of the numbering when it is sharded: each of the
                            spf_raster_pair_shards(S, G) shards owns pair_capacity / shards records (see there) */
    int64_t raw_stride;  /* sh_layout 3 only: floats between two Gaussians' rows of SpfInputs.raw (>= 7 + 3K) */
    float adapter_eps;   /* sh_layout 3 only: the eps of rotations = q / (|q| + eps) (gaussian_adapter.py:136) */
} SpfDims;

/* Inputs (all float32, contiguous, row-major). */
typedef struct SpfInputs {
    const float* means3D;    /* [S,G,3] */
    const float* scales;     /* [S,G,3] */
    const float* rotations;  /* [S,G,4] quaternion (r,x,y,z), used as given (not normalised) */
    const float* opacities;  /* [S,G]   */
    const float* shs;        /* [S,G,K,3] ([S,G,3,K] with sh_layout 1; [S,G,3,16] with sh_layout 2) or NULL when colors is set */
    const float* colors;     /* [S,G,3]  or NULL when shs is set (colors_precomp) */
    const float* viewmatrix; /* [S,V,4,4] world->view, row-vector convention (p_view = [p,1] @ M) */
    const float* projmatrix; /* [S,V,4,4] perspective only, row-vector convention */
    const float* tanfov;     /* [S,V,2] (tanfovx, tanfovy) */
    const float* bg;         /* [S,V,3] background colour */
    const float* view_scale; /* [S,V] or NULL (= 1): per-render world scale applied to means3D and scales
                                inside the projection kernel -- the reference's scale-invariant
                                normalisation (cuda_splatting.py:66-74) without per-view copies */
    const double* viewmatrix64; /* [S,V,4,4] or NULL: the same world->view matrix in float64 with view_scale already
                                folded into its first three rows, so that [p,1] @ M64 is the (rescaled) view-space
                                position of the UNscaled mean p.  Written by spf_camera_forward / spf_decoder_prepare.
                                The view-space position p R + t is a difference of terms ~|t| for Gaussians near the
                                camera (after the 1/near rescale |t| is tens of units while z may be 0.2): the
                                projection kernels form it in float64 -- from this matrix when given, else from the
                                float32 one promoted -- and round once.  Everything else stays float32. */
    const float* shs_high;   /* sh_layout 2 only: [S,G,3,9], band 4; read only when sh_band4 is set (may be NULL otherwise) */
    const float* raw;        /* sh_layout 3 only: [S*G rows of 7+3K floats at SpfDims.raw_stride]: the adapter's input, read in
                                place (the encoder hands over gaussians[..., 1:], a view of its 83-channel head output) */
    const float* sh_mask;    /* sh_layout 3 only: [K] per-coefficient scale of the harmonics (gaussian_adapter.py:47-48) */
} SpfInputs;

/* State written by the forward pass and read by the backward pass (owned by the caller, e.g. the
 * autograd context).  T = ceil(W/16)*ceil(H/16) tiles per render, P = H*W pixels per render. */
typedef struct SpfState {
    float* rec;            /* [R*G,12]  screen-space record: xy, conic(3), opacity, rgb, depth, cull r^2, flags */
    int32_t* radii;        /* [R*G]     pixel radius, 0 = culled (also an output) */
    uint32_t* rect;        /* [R*G]     packed tile rect: xmin | ymin<<8 | xmax<<16 | ymax<<24 */
    float* zkey;           /* [R*G]     view-space depth again, compact (the binning pass reads 8 B per Gaussian
                                        instead of pulling the 48-byte record through the cache) */
    uint32_t* tile_count;  /* [R*T]     Gaussians per tile */
    uint32_t* tile_start;  /* [R*T+1]   exclusive scan of tile_count; last = D.  Direct bins: scratch -- when tile_start |
                            *            tile_fill are ONE 8-byte aligned piece (tile_fill == tile_start + R*T + 1, as in the
                            *            one-buffer layout below) and the call has >= 2,048 tiles, the library keeps the
                            *            composite kernels' launch order there: [R*T][2] = (tile | dense-backward << 31 | dense-forward << 30, list length),
                            *            longest lists first, written by spf_raster_forward_render and read again by
                            *            spf_raster_backward -- leave it alone between the two */
    uint32_t* tile_fill;   /* [R*T]     scratch cursor for the binning pass (direct bins: see tile_start) */
    uint32_t* tile_flags;  /* [R*T]     footprint load of the tile: sum over its list of min(cull-disc bounding-box
                                        area in pixels, 256); tiles whose mean exceeds SPF_DENSE_AREA(_FWD) take the dense
                                        "rows" form of the compositing kernels, the others the sparse "lists" form */
    uint32_t* counters;    /* [4]       0: D (total pairs) 1: max tile_count 2: plan verdict (0 = held) 3: number of dense tiles
                                        (direct bins: 0 and 3 are not maintained) */
    uint64_t* pairs;       /* [capacity] per-tile lists, each sorted by (depth bits << 32 | Gaussian id)
                                        (direct bins: [R*T*bin_cap], tile t's list at t * bin_cap) */
    uint32_t* pair_off;    /* [R*G,2]   (packed tile rect as in `rect`, index of the Gaussian's first (Gaussian, tile) pair):
                                        its pair with the k-th tile of its rect (row-major) has index pair_off + k.  One
                                        8-byte record so that the composite backward finds a list entry's gradient slot
                                        with ONE gather (round 3: rect and pair_off were two arrays, two gathers) */
    uint32_t* blk_total;   /* [R*nblk]  pairs per block of 256 Gaussians, nblk = spf_raster_view_partial_blocks(G) */
    uint32_t* blk_base;    /* [R*nblk]  exclusive scan of blk_total */
    float* final_T;        /* [R*P]     transmittance left after the last contributor */
    uint32_t* n_contrib;   /* [R*P]     per pixel: 1 + list position of the last contributor (0 = none) */
    uint32_t* pair_cursor; /* [8]       direct bins only (may be NULL otherwise): cursors of the pair numbering, ZERO on
                                        entry of spf_raster_forward_project* (spf_decoder_prepare clears them when they
                                        lie inside the buffer it is given) */
    uint8_t* sh_clamp;     /* [R*G]     SH colours only (may be NULL with colors_precomp): bit c = colour channel c of this
                                        (render, Gaussian) was clamped at 0 by the forward (such a
                                        channel passes no gradient).  Written by spf_raster_forward_project*, read by
                                        spf_raster_backward for sh_degree >= 1 -- the backward contracts dL/dcolour with the
                                        coefficients BEFORE the basis derivatives (three accumulators instead of twelve: the
                                        degree-4 kernel fits two waves per SIMD) and so needs the clamp decision up front
                                        instead of re-evaluating the colour */
    uint32_t* verdict_host; /* [1]      may be NULL.  A HOST-MAPPED word (hipHostMalloc / pinned memory, device-accessible) that
                                        the projection kernel of a direct-bins call stores a non-zero value to when it raises
                                        a plan flag (counters[2]); the caller zeroes it before the call.  A host that wants the
                                        verdict EARLY then needs no device->host copy on the stream (a 4-byte copy is a trip
                                        through the copy engine that the next kernel of the stream waits for): it records
                                        an event behind spf_raster_forward_project*, queues the rest of the chain, waits for the
                                        event and reads the word */
} SpfState;

typedef struct SpfOutputs {
    float* image;  /* [R,3,H,W] */
    float* depth;  /* [R,1,H,W]  sum_i z_i alpha_i T_i */
    float* alpha;  /* [R,1,H,W]  1 - final_T */
} SpfOutputs;

/* Upstream gradients (any may be NULL = zero) and the gradients to produce (any may be NULL = skip). */
typedef struct SpfGrads {
    const float* dL_dimage;   /* [R,3,H,W] */
    const float* dL_ddepth;   /* [R,1,H,W] */
    const float* dL_dalpha;   /* [R,1,H,W] */
    float* gpair;             /* [capacity,10] scratch: screen-space gradient of every (Gaussian, tile) pair, written
                                 once per pair by its tile (no global atomics, no memset), indexed by pair_off + k;
                                 records are packed: 9 floats each, 10 when dL_ddepth != NULL
                                 (dL/d pixel centre xy, dL/d 2-D covariance (a, b, c), dL/d opacity, dL/d rgb[, dL/d depth]) */
    float* vpartial;          /* [R, nblk, 12] (16-byte aligned) scratch for the deterministic viewmatrix reduction,
                                 nblk = spf_raster_view_partial_blocks(G) */
    float* dL_dmeans3D;       /* [S,G,3] */
    float* dL_dscales;        /* [S,G,3]   (NULL when enable_cov_grad is false) */
    float* dL_drotations;     /* [S,G,4]   (NULL when enable_cov_grad is false) */
    float* dL_dopacities;     /* [S,G]   */
    float* dL_dshs;           /* same layout as shs (NULL when enable_sh_grad is false) */
    float* dL_dcolors;        /* [S,G,3]   */
    float* dL_dviewmatrix;    /* [S,V,4,4] */
    float* dL_dmeans2D;       /* [R,G,3]   NDC-scaled screen-space gradient (xy, 0) */
    float* dL_dshs_high;      /* sh_layout 2 with sh_band4 only: [S,G,3,9] (otherwise never touched, may be NULL: the
                                 gradient of band 4 is zero and its consumer, spf_adapter_backward, takes NULL for that) */
    float* dL_draw;           /* sh_layout 3 only: [S*G, 7+3K] contiguous -- what spf_adapter_backward would have written;
                                 dL_dscales / dL_drotations / dL_dshs are not used then */
} SpfGrads;

/* Camera set-up for R = S*V renders (all float32, contiguous). */
typedef struct SpfCamera {
    const float* extrinsics;  /* [R,4,4] camera-to-world (OpenCV), as the decoder receives it */
    const float* intrinsics;  /* [R,3,3] normalised */
    const float* near;        /* [R] */
    const float* far;         /* [R] */
    float* viewmatrix;        /* [R,4,4] out: inverse(extrinsics')^T (row-vector convention) */
    float* projmatrix;        /* [R,4,4] out: perspective^T */
    float* tanfov;            /* [R,2]   out */
    float* view_scale;        /* [R]     out (may be NULL): 1/near when scale_invariant else 1 */
    int32_t R;
    int32_t scale_invariant;  /* cuda_splatting.py:66-74: translation, means, scales x 1/near; near -> 1 */
    double* viewmatrix64;     /* [R,4,4] out (may be NULL): see SpfInputs.viewmatrix64 */
} SpfCamera;

int spf_abi_version(void);
const char* spf_last_error(void);

/* Number of tiles per render and size of the vpartial scratch. */
int spf_raster_num_tiles(int32_t H, int32_t W);
int spf_raster_view_partial_blocks(int32_t G);
/* Which tile (render * T + tile) block slot `slot` (< R*T / 8) of XCD `xcd` (0..7) of a composite lists launch stands for
 * before the end of every XCD's range is sorted longest list first (planned calls on direct bins of >= 2,048 tiles, R*T
 * a multiple of 8): a contiguous range of renders per XCD -- or, for calls of exactly eight renders, strips of 64 tiles
 * dealt out so that no XCD is left with ONE render (strip q of render r -> XCD (r + q) % 8).  A bijection; host-side
 * arithmetic only (documentation and tests).  -1 for arguments out of range. */
int spf_raster_launch_slot_tile(int32_t R, int32_t T, int32_t xcd, int32_t slot);
/* Direct bins: how the (Gaussian, tile) pair numbering of a call of S scenes x G Gaussians is sharded, and the largest
 * number of tiles per render whose histogram the projection kernel keeps in LDS (direct bins need it).
 * spf_raster_pair_shards: 1 or 8.  With 8 shards the blocks of the projection kernel number their pairs from eight
 * cursors (block b -> shard b % 8; one cursor would be a hot word) and shard i owns the gradient records
 * [i * pair_capacity / 8, (i + 1) * pair_capacity / 8): plan flag 1 is raised when ONE shard outgrows its eighth, so a
 * caller that wants "D <= capacity never fails" sizes pair_capacity (and g->gpair) with headroom for the imbalance of a
 * round-robin deal of blocks -- the Python host passes 1.25 x the planned capacity (the shards are interleaved samples of
 * the same scenes: they differ by per cents). */
int spf_raster_pair_shards(int32_t S, int32_t G);
int spf_raster_max_lds_tiles(void);
/* Into how many chunks of renders spf_raster_forward_render (backward = 0) / spf_raster_backward (backward = 1) split a
 * call of S scenes x V views.  1 unless the environment says SPF_CHUNKS=n: then, after the joint tile scan, the chunks
 * (whole scenes each) run as independent launch chains alternating between the caller's stream and one auxiliary stream
 * of the library (fork / join by events; capturable in a HIP graph); results are bit-identical to the single chain and
 * the stage timing below counts one launch per chunk.  Measured slower than the single chain on MI355X (see api.hip),
 * hence off by default. */
int spf_raster_chunks(int32_t S, int32_t V, int32_t H, int32_t W, int32_t backward);

/* Camera tensors from poses / intrinsics, and the gradient of the poses from dL/dviewmatrix
 * (dL_dviewmatrix [R,4,4] in, dL_dextrinsics [R,4,4] out; cam->viewmatrix must hold the forward result). */
int spf_camera_forward(const SpfCamera* cam, void* stream);
int spf_camera_backward(const SpfCamera* cam, const float* dL_dviewmatrix, float* dL_dextrinsics, void* stream);

/* Forward, stage 1: per-Gaussian projection / 2-D covariance / colour, per-tile counts and their
 * scan.  On return (stream order) st->counters[0] = D and st->counters[1] = longest tile list. */
int spf_raster_forward_project(const SpfDims* d, const SpfInputs* in, SpfState* st, void* stream);

/* Decoder fast path (the batched DecoderSplattingCUDA.forward, decoder_splatting_cuda.py:41-78): two launches fewer
 * per step than the calls above.
 *   spf_decoder_prepare           = spf_camera_forward AND the clearing of `zero_bytes` bytes at `zero` in ONE kernel.
 *                                   Pass ONE buffer laid out tile_count | tile_flags | tile_start (R*T+1) | tile_fill |
 *                                   counters (4) [| pair_cursor (8) | padding] and clear all of it (16*R*T + 20 bytes,
 *                                   + 32 with the cursors, rounded up to 16; 16-byte aligned);
 *   spf_raster_forward_project_prepared = spf_raster_forward_project without its own clearing.  `cleared_bytes` says how
 *                                   much of that buffer (from tile_count on) the caller cleared: all of it -> the tile scan
 *                                   runs with one block per render instead of a single block; only the first 8*R*T bytes
 *                                   (the two count arrays) or a buffer laid out differently -> the self-initialising
 *                                   single-block scan; less -> the call clears the counts itself.  Nothing is assumed;
 *   spf_camera_backward_partials  = the deterministic sum of g->vpartial [R,nblk,12] (as written by spf_raster_backward
 *                                   when g->vpartial != NULL; pass g->dL_dviewmatrix = NULL to skip its own reduction)
 *                                   AND spf_camera_backward, in ONE kernel. */
int spf_decoder_prepare(const SpfCamera* cam, void* zero, uint64_t zero_bytes, void* stream);
int spf_raster_forward_project_prepared(const SpfDims* d, const SpfInputs* in, SpfState* st, uint64_t cleared_bytes,
                                        void* stream);
int spf_camera_backward_partials(const SpfCamera* cam, const float* vpartial, int32_t nblk, float* dL_dextrinsics,
                                 void* stream);

/* Forward, stage 2: bin (Gaussian, tile) pairs into per-tile lists, depth-sort every list and
 * composite.  `capacity` = number of uint64 entries st->pairs can hold.  `max_tile_hint` = upper bound of the
 * longest tile list the caller assumes (exact mode: host copy of counters[1]; 0 = unknown: every sort size class
 * is launched).  `dense_tiles_hint` is IGNORED (pass SPF_UNKNOWN): up to round 4 sparse and dense tiles had a kernel
 * each and the hint let the host skip one; since round 5 one kernel composites every tile in the form that suits it.
 * A caller that PLANS the call from an earlier one instead of reading the counters back (no device->host
 * synchronisation; capturable in a HIP graph) passes its assumptions here and they are checked on the device:
 * counters[2] = 0 if the plan held, else a bit mask: 1 = D > capacity, 2 = a tile list longer than max_tile_hint
 * (it could not be sorted); bit 4 (a wrong dense_tiles_hint, rounds 2 - 4) is never set.  A failed plan is
 * never silent and never undefined: with a non-zero flag EVERY output of this call (image, depth, alpha) and every
 * gradient of the matching backward is filled with NaN -- deterministic, and caught by the reference's own
 * NaN-gradient guard (src/model/model_wrapper.py:1117-1151) even if the caller never reads the flag. */
int spf_raster_forward_render(const SpfDims* d, const SpfInputs* in, SpfState* st, SpfOutputs* out,
                              uint64_t capacity, uint32_t max_tile_hint, uint32_t dense_tiles_hint, void* stream);

/* Backward of both stages.  `capacity` = number of 10-float records g->gpair can hold (>= D); `dense_tiles_hint`: ignored,
 * as in spf_raster_forward_render. */
int spf_raster_backward(const SpfDims* d, const SpfInputs* in, const SpfState* st,
                        const SpfGrads* g, uint64_t capacity, uint32_t dense_tiles_hint, void* stream);

/* Fused Gaussian adapter (UnifiedGaussianAdapter.forward, src/model/encoder/common/gaussian_adapter.py:122-150):
 * raw[N, 7+3K] network channels (row stride `raw_stride` floats >= 7+3K: the encoder hands over `gaussians[..., 1:]`, a view
 * of its 83-channel head output, encoder_spfsplatv2.py:261-268 -- read in place, no contiguous copy) ->
 * scales[N,3] = min(0.001*softplus, 0.3), rotations[N,4] = q/(|q|+eps), harmonics[N,3,K] = raw[7:] * sh_mask[K];
 * with harmonics_high != NULL (K = 25): the band-split layout, harmonics = [N,3,16] and harmonics_high = [N,3,9]
 * (SpfDims.sh_layout 2).  One pass: every raw row is read once, through LDS, and every output is written coalesced.
 * Backward: dL_draw[N, 7+3K] (contiguous); any upstream gradient may be NULL = zero; `split` says dL_dharmonics is
 * [N,3,16] with band 4's gradient in dL_dharmonics_high [N,3,9] -- or NULL, which is never read and costs nothing. */
int spf_adapter_forward(const float* raw, int64_t raw_stride, int64_t N, int32_t K, const float* sh_mask, float eps,
                        float* scales, float* rotations, float* harmonics, float* harmonics_high, void* stream);
int spf_adapter_backward(const float* raw, int64_t raw_stride, int64_t N, int32_t K, const float* sh_mask, float eps,
                         const float* dL_dscales, const float* dL_drotations, const float* dL_dharmonics,
                         const float* dL_dharmonics_high, int32_t split, float* dL_draw, void* stream);

/* Photometric MSE on the decoder output (LossMse.forward, src/loss/loss_mse.py:36-51):
 * loss[0] = weight * mean((prediction - image)^2) over n floats, and its backward
 * dL_dprediction = (2 * weight / n) * dL_dloss[0] * (prediction - image) (dL_dloss is read on the device).
 * partial: scratch of spf_mse_partial_blocks() floats.  The sum is taken in a fixed order: results are run-to-run
 * identical.  Tensors 16-byte aligned. */
int spf_mse_partial_blocks(void);
int spf_mse_forward(const float* prediction, const float* image, int64_t n, float weight, float* partial,
                    float* loss, void* stream);
int spf_mse_backward(const float* prediction, const float* image, int64_t n, float weight, const float* dL_dloss,
                     float* dL_dprediction, void* stream);
/* The same forward that ALSO writes dL_dprediction_unit[i] = (2 * weight / n) * (prediction[i] - image[i]) -- the gradient
 * for dL_dloss = 1 -- so that the backward is spf_mse_scale_grad: dL_dprediction[i] *= dL_dloss[0] in place, which returns
 * after one scalar read when dL_dloss[0] is exactly 1 (what `loss.backward()` passes): the backward of the loss costs a
 * launch instead of a pass over prediction, image and gradient. */
int spf_mse_forward_grad(const float* prediction, const float* image, int64_t n, float weight, float* partial,
                         float* loss, float* dL_dprediction_unit, void* stream);
int spf_mse_scale_grad(float* dL_dprediction, int64_t n, const float* dL_dloss, void* stream);

/* In-place 2-D rotary embedding.  tokens[B,N,H,D] with element strides (stride_b, stride_n, stride_h) and
 * stride(D) == 1; dtype: 0 = float32, 1 = float16, 2 = bfloat16.  positions[B / pos_div, N, 2] int64 contiguous
 * (y, x): batch item b uses positions[b / pos_div] (pos_div = 1 for the CroCo layout; a head-major [B,H,N,D] tensor
 * is passed as B*H batches of one head with pos_div = H).  fwd = +F0 for forward, -F0 for backward. */
int spf_rope2d(void* tokens, const int64_t* positions, int32_t B, int32_t N, int32_t H, int32_t D,
               int64_t stride_b, int64_t stride_n, int64_t stride_h, int32_t pos_div, int32_t dtype, float base,
               float fwd, void* stream);
/* The same rotation applied to TWO tensors of identical shape, strides, dtype and positions in one launch (the angles
 * are evaluated once): q and k of an attention layer, which the reference rotates with two calls
 * (src/model/encoder/backbone/croco/blocks.py:102-104). */
int spf_rope2d_pair(void* tokens, void* tokens2, const int64_t* positions, int32_t B, int32_t N, int32_t H, int32_t D,
                    int64_t stride_b, int64_t stride_n, int64_t stride_h, int32_t pos_div, int32_t dtype, float base,
                    float fwd, void* stream);

/* Per-stage device timing with HIP events recorded on the launch stream around every kernel
 * stage.  spf_stage_timing_enable(mask) clears the log and starts recording the stages whose bit
 * (1 << SPF_STAGE_x) is set in `mask` (0 = off, -1 = all) (up to
 * SPF_STAGE_LOG launches per stage are kept); spf_stage_times_ms synchronises on the recorded
 * events and returns, per stage, the summed device time and the number of launches logged since
 * enable -- average launch duration = total_ms[i] / count[i]. */
enum {
    SPF_STAGE_PROJECT = 0,   /* forward per-Gaussian kernel */
    SPF_STAGE_SCAN = 1,
    SPF_STAGE_BIN = 2,
    SPF_STAGE_SORT = 3,
    SPF_STAGE_RENDER_FWD = 4,
    SPF_STAGE_RENDER_BWD = 5,
    SPF_STAGE_PROJECT_BWD = 6,
    SPF_STAGE_ROPE = 7,
    SPF_STAGE_COUNT = 8
};
#define SPF_STAGE_LOG 1024
int spf_stage_timing_enable(int32_t mask);
/* Record only every n-th launch of an enabled stage (default 1): an event pair leaves the GPU idle for ~11 us. */
int spf_stage_timing_sample_every(int32_t n);
int spf_stage_times_ms(float* total_ms /* host, [SPF_STAGE_COUNT] */,
                       int32_t* count /* host, [SPF_STAGE_COUNT] */);
const char* spf_stage_kernel_name(int32_t stage);

#ifdef __cplusplus
}
#endif
#endif /* SPFSPLAT_HIP_H */
