/*
 * sv_hip.h — C-ABI of libsvhip.so, the MI355X (gfx950) sparse-voxel inference library.
 *
 * This is the drop-in boundary for the hot path of bcsefercik/markerless-robot-camera-calibration
 * (SURVEY.md §8b): everything the reference gets from MinkowskiEngine 0.5.4 / numpy LAPACK on the
 * path  voxelise -> sparse U-Net -> slice/argmax -> Kabsch  is exported here as plain C functions.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (memory owned by the caller, normally a PyTorch-ROCm tensor)
 *    unless the name ends in _host;
 *  - `stream` is a hipStream_t passed as void*; NULL = the default stream;
 *  - the library allocates nothing persistent: temporaries come from a caller-supplied workspace whose
 *    size is returned by the matching *_workspace_bytes();
 *  - return value: 0 = OK, <0 = error (SV_ERR_*); sv_last_error() gives the message (thread-local);
 *  - dynamic sizes (number of voxels ...) are written to a small device `counters` array the caller
 *    reads back (one D2H copy), never returned through host pointers, so calls stay asynchronous.
 *
 * Row order of every coordinate map is CANONICAL: ascending 64-bit key
 *    key = batch << 54 | morton3(x + 2^17, y + 2^17, z + 2^17)       (x in bit 3j, y in 3j+1, z in 3j+2)
 * so a stride-2 parent key is the child key with 3 low Morton bits cleared and children of one parent are
 * contiguous rows.  (MinkowskiEngine leaves row order unspecified — SURVEY.md Appendix B.2.)
 *
 * Kernel-offset numbering (index into W[K][Cin][Cout]):
 *    kernel_size 3:  k = (dx+1) + 3*(dy+1) + 9*(dz+1),  dx,dy,dz in {-1,0,1}   (x fastest)
 *    kernel_size 2:  k = dx + 2*dy + 4*dz,              dx,dy,dz in {0,1}
 *    transposed kernel_size 2 stride 2: the fine voxel c with parent p uses k of (c - p) / tensor_stride.
 */
#ifndef SV_HIP_H
#define SV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_OK 0
#define SV_ERR_INVALID (-1)   /* bad shape / null pointer / unsupported parameter */
#define SV_ERR_WORKSPACE (-2) /* workspace too small */
#define SV_ERR_HIP (-3)       /* HIP runtime error, see sv_last_error() */
#define SV_ERR_RANGE (-4)     /* coordinate or batch index outside the key range (reported in counters) */

#define SV_ACT_NONE 0
#define SV_ACT_RELU 1
#define SV_ACT_LEAKY_RELU 2

#define SV_POOL_MAX 0
#define SV_POOL_AVG 1

#define SV_REDUCE_MEAN 0  /* ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE */
#define SV_REDUCE_FIRST 1 /* ME.utils.sparse_quantize: lowest original index represents the voxel */

#define SV_TILE_ROWS 128 /* output rows per conv tile; plans are padded to a multiple of this */
#define SV_COORD_BIAS 131072 /* 2^17 */
#define SV_COORD_BITS 18
#define SV_MAX_BATCH 1024

typedef void* sv_stream_t;

const char* sv_last_error(void);
/* 2: sv_conv_fwd takes V_in (rows of `in`); sv_single_linkage_roots / sv_select_equal added
 * 3: sv_plan_build takes nbr_base (plans of a batch range of a kernel map); sv_key_point_predictions,
 *    sv_conv_last_instance and sv_conv_fwd_acc (offset-range passes of one layer) added
 * 4: sv_conv_set_dispatch (per-thread dispatch thresholds: one frame alone vs frames overlapped), the frame composites
 *    sv_frame_maps / sv_frame_plans (a frame's coordinate work as two host calls), sv_topk_indices (get_pred_center),
 *    sv_key_point_predictions_batched */
#define SV_ABI_VERSION 4
int sv_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * A1/A2  voxelisation   (replaces ME.TensorField(...).sparse(), ME.SparseTensor(coordinates=...),
 *        ME.utils.sparse_quantize; reference call sites app/inference_engine.py:405-415,446-454,540-549,
 *        test_segmentation.py:62-70, data/alivev2.py:290-296, train_segmentation.py:78)
 *
 * coords4: float32[N,4] rows (batch, x, y, z) already multiplied by `scale` by the caller, exactly what
 *          ME.utils.batched_coordinates hands to TensorField.  voxel = floor(coord) per axis.
 *          With coords_are_int != 0 the buffer is int32[N,4] (already quantised coordinates).
 * keys:    uint64[N]    first V entries = canonical sorted unique keys
 * vcoords: int32[N,4]   first V rows   = (batch,x,y,z) of each voxel
 * inverse: int64[N]     voxel row of every input point  (TensorField -> SparseTensor inverse map)
 * order:   int32[N]     point indices sorted by (key, original index)  (stable)
 * seg_start:int32[N+1]  first V+1 entries: voxel v owns order[seg_start[v] .. seg_start[v+1])
 * counters:int32[4]     [0] = V, [1] = number of out-of-range points (must be 0), [2..3] reserved
 * ------------------------------------------------------------------------------------------- */
size_t sv_voxelize_workspace_bytes(int64_t N);
int sv_voxelize(const void* coords4, int coords_are_int, int64_t N, void* workspace, size_t workspace_bytes,
                uint64_t* keys, int32_t* vcoords, int64_t* inverse, int32_t* order, int32_t* seg_start,
                int32_t* counters, sv_stream_t stream);

/* per-voxel feature reduction: out[v][c] = mean (or first) of feats[order[j]][c], j in the voxel's segment,
 * summed sequentially in ascending original point index (deterministic). */
int sv_voxel_reduce(const float* feats, int C, const int32_t* order, const int32_t* seg_start, int64_t V, int mode,
                    float* out, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Coordinate manager pieces (replace ME's coordinate_map_gpu / kernel_map; implicit in every
 * ME.MinkowskiConvolution call of model/backbone/minkunet.py:55-121)
 * ------------------------------------------------------------------------------------------- */
/* open-addressing hash  key -> row ;  capacity must be a power of two >= 2*V. */
int sv_hash_build(const uint64_t* keys, int64_t V, uint64_t* table_keys, int32_t* table_vals, int64_t capacity,
                  sv_stream_t stream);

/* stride-2 coordinate map of a map at tensor stride 2^level:  out coord = floor(c / 2^(level+1)) * 2^(level+1).
 * parent: int32[V_in] row of each input voxel in the output map.  child_start: int32[V_in+1] (first V_out+1 used).
 * counters[0] = V_out. */
size_t sv_stride_map_workspace_bytes(int64_t V_in);
int sv_stride_map(const uint64_t* keys_in, int64_t V_in, int level, void* workspace, size_t workspace_bytes,
                  uint64_t* keys_out, int32_t* vcoords_out, int32_t* parent, int32_t* child_start, int32_t* counters,
                  sv_stream_t stream);

/* 27-offset neighbour table of a map with itself (kernel_size 3, stride 1):
 * nbr[k*ld + o] = row of voxel at coords[o] + offset_k * tensor_stride * dilation, or -1.   mask[o] bit k = present. */
int sv_kernel_map_k3(const int32_t* vcoords, int64_t V, int tensor_stride, int dilation, const uint64_t* table_keys,
                     const int32_t* table_vals, int64_t capacity, int32_t* nbr, int64_t ld, uint32_t* mask,
                     sv_stream_t stream);
/* kernel_size 2 stride 2 (down): out rows = coarse voxels; nbr[k*ld + p] = fine row of child k of p, or -1. */
int sv_kernel_map_down(const uint64_t* keys_fine, const int32_t* parent, int64_t V_fine, int level, int64_t V_coarse,
                       int32_t* nbr, int64_t ld, uint32_t* mask, sv_stream_t stream);
/* transposed kernel_size 2 stride 2 (up): out rows = fine voxels; nbr[k*ld + i] = parent[i] iff k == child id of i. */
int sv_kernel_map_up(const uint64_t* keys_fine, const int32_t* parent, int64_t V_fine, int level, int32_t* nbr,
                     int64_t ld, uint32_t* mask, sv_stream_t stream);

/* Conv execution plan: rows sorted by neighbour mask so that 16-row MFMA sub-tiles share offsets.
 * perm:    int32[Vpad]        output row handled at sorted position r (-1 = padding)
 * nbr_s:   int32[K][Vpad]     nbr[k][perm[r]]
 * submask: uint32[Vpad/128][K] bit s set = sub-tile s (rows 16s..16s+15 of the tile) has a neighbour at offset k
 * tile_order: int32[Vpad/128] plan tiles sorted by work (number of active (offset, sub-tile) slots) descending: the
 *          conv kernel dispatches its workgroups in this order (longest first) so the launch has a short tail
 * Vpad = round_up(V, 128).
 * Plans of a ROW RANGE of a kernel map (batched tensors whose feature tables exceed the 2 GB extent of the buffer-addressed
 * conv instances: the frames of a batch never share neighbours, data/alivev2.py:358-383): pass nbr + o0 / mask + o0 / V = o1 - o0
 * for output rows [o0, o1) and nbr_base = first input row of the range; every stored index is then relative to that row, and
 * sv_conv_fwd runs on `in + nbr_base * in_ld`, `out + o0 * out_ld`.  nbr_base = 0 for a whole map.
 * `perm` (and every plan array) must be 16-byte aligned: the conv epilogue reads it four entries at a time. */
size_t sv_plan_workspace_bytes(int64_t V);
int sv_plan_build(const int32_t* nbr, int64_t ld, const uint32_t* mask, int K, int64_t V, int64_t nbr_base, void* workspace,
                  size_t workspace_bytes, int32_t* perm, int32_t* nbr_s, uint32_t* submask, int32_t* tile_order,
                  int64_t Vpad, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Frame composites: the coordinate work of ONE frame as two host calls (the reference reaches all of it implicitly through
 * ME.TensorField(...).sparse(), app/inference_engine.py:405-415, and the ME.MinkowskiConvolution calls of
 * model/backbone/minkunet.py:125-183; its consumer calls InferenceEngine.predict once per frame, app/main.py:432-456, so the
 * host time of ~200 launches behind ~120 calls is frame latency).  Same kernels, same array contents as the piecewise entry
 * points above.
 *
 * sv_frame_maps: voxelise (as sv_voxelize) + `levels` stride-2 maps (as sv_stride_map).  The level sizes are read back
 *   inside the call through `counters_host` (PINNED host memory, 4 * (levels + 2) int32): the only host synchronisations of
 *   a frame's coordinate work, each waiting for this call's own kernels on `stream`.  Outputs are carved from `arena`
 *   (sv_frame_maps_arena_bytes: worst case V_l <= N); `scratch` (sv_frame_maps_scratch_bytes) is free again on return.
 *   layout (host int64[8 + 6 * (levels + 1)]): [0] arena bytes used, [1] N, [2] levels, [3] offset of inverse int64[N],
 *   [4] order int32[N], [5] seg_start int32[V_0 + 1], [6] points outside the key range (then SV_ERR_RANGE), then per level l
 *   six entries: V_l, offset of keys uint64[V_l], vcoords int32[V_l][4], parent int32[V_l] (row of each voxel in level
 *   l + 1; -1 at the last level), child_start int32[V_{l+1} + 1] (-1 at the last level), 0.
 * sv_frame_plans: hash tables, kernel maps and conv plans of levels 0..levels; no synchronisation.  flags select what is
 *   built: SV_FRAME_K3 (27-offset map + plan per level), SV_FRAME_DOWN / SV_FRAME_UP (the kernel_size 2 stride 2 maps
 *   between levels l and l + 1 and their plans), SV_FRAME_SPLIT (offset-range plans: split_cuts[l][SV_FRAME_MAX_CUTS] holds
 *   level l's ascending split points, 0-terminated; they need the level's 27-offset map - built by SV_FRAME_K3 in the same
 *   call or passed in k3_nbr[l] / k3_mask[l], so that a caller can build them in a second call while the encoder runs).
 *   keys / coords / parent: per-level device pointers (parent[l] as in sv_frame_maps).  layout (host int64[16 * (1 +
 *   max_records)]): [0] arena bytes used, [1] number of records; record r at 16 * (1 + r): kind (SV_FRAME_REC_*), level, k0,
 *   k1, offset of the raw map int32[K][ld] (hash: of the table values; -1: rows k0..k1 of the caller's k3_nbr), of its mask
 *   (split: the range's mask), of perm, nbr_s, submask, tile_order, V_out, Vpad, K, ld, offset of the hash keys, capacity.
 * ------------------------------------------------------------------------------------------- */
#define SV_FRAME_MAX_LEVELS 8
#define SV_FRAME_MAX_CUTS 4
#define SV_FRAME_K3 1
#define SV_FRAME_DOWN 2
#define SV_FRAME_UP 4
#define SV_FRAME_SPLIT 8
#define SV_FRAME_RECORD 16
#define SV_FRAME_REC_HASH 1
#define SV_FRAME_REC_K3 2
#define SV_FRAME_REC_DOWN 3
#define SV_FRAME_REC_UP 4
#define SV_FRAME_REC_SPLIT 5
size_t sv_frame_maps_arena_bytes(int64_t N, int levels);
size_t sv_frame_maps_scratch_bytes(int64_t N);
int sv_frame_maps(const void* coords4, int coords_are_int, int64_t N, int levels, void* arena, size_t arena_bytes, void* scratch,
                  size_t scratch_bytes, int32_t* counters_host, int64_t* layout, sv_stream_t stream);
size_t sv_frame_plans_arena_bytes(const int64_t* V, int levels, int flags, const int32_t* split_cuts);
size_t sv_frame_plans_scratch_bytes(const int64_t* V, int levels);
int sv_frame_plans(const void* const* keys, const void* const* coords, const void* const* parent, const int64_t* V, int levels,
                   int flags, const int32_t* split_cuts, const void* const* k3_nbr, const void* const* k3_mask, void* arena,
                   size_t arena_bytes, void* scratch, size_t scratch_bytes, int64_t* layout, int max_records, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A3/A5  sparse convolution, output stationary, fp32 MFMA, fused epilogue
 *   (replaces ME.MinkowskiConvolution / ConvolutionTranspose / Linear + MinkowskiBatchNorm(eval) +
 *    residual add + ReLU/LeakyReLU: model/backbone/minkunet.py:125-187, resnet.py:95-127,
 *    model/robotnet_segmentation.py:55-64)
 *
 *   acc[o][n] = sum over k ascending, c ascending of  in[nbr[k][o]][c] * W[k][c][n]     (one fmaf chain)
 *   y = acc * scale[n] + shift[n]  (fmaf; scale NULL = 1, shift NULL = 0)   BN(eval) folded / bias
 *   y += residual[o][n]            (residual NULL = none)
 *   out[o][n] = act(y)
 * perm/nbr_s/submask NULL = dense rows (kernel_size 1 / Linear): nbr = identity.  tile_order may be NULL.
 * V_in = rows of `in` (every index in nbr_s is below it; = V_out for dense rows): with it the wide-layer kernels
 * address `in` through a bounds-checked buffer descriptor (32-bit offsets, absent neighbours read as zero rows);
 * inputs of 2 GB and more take the guarded form with 64-bit addresses - except dense rows (no plan, K = 1), which this
 * entry point splits into row ranges below the extent, and batched tensors, whose callers launch one batch range at a
 * time with plans built for that range (sv_plan_build: nbr_base).
 * One entry point, several kernels behind it (all with the chain order above, so results do not depend on the
 * choice): fp32-MFMA tiles for the wide layers, their fused-offset form for 32/64-channel inputs, a thread-per-voxel
 * VALU kernel for the 3-channel first layer and a row-streaming VALU kernel for dense layers with <= 4 outputs.
 * ------------------------------------------------------------------------------------------- */
int sv_conv_fwd(const float* in, int64_t V_in, int64_t in_ld, int Cin, const float* W, int K, int Cout, const int32_t* perm,
                const int32_t* nbr_s, const uint32_t* submask, const int32_t* tile_order, int64_t V_out, int64_t Vpad,
                const float* scale,
                const float* shift, const float* residual, int64_t res_ld, int act, float slope, float* out,
                int64_t out_ld, sv_stream_t stream);
/* The same layer with its chains CONTINUED from an earlier launch: acc_init[o][n] (row stride acc_ld, NULL = start at 0) is
 * the raw accumulator of output element (o, n) over the kernel offsets that precede this launch's - the caller splits the K
 * offsets of a layer into ascending ranges, runs every range with its own plan (its own row order: rows that share their
 * neighbours among 13-14 offsets group far better into 16-row matrix-op sub-tiles than rows that must share all 27:
 * 0.87 -> 0.96 useful row slots on the 2 cm room level) and weight block W + k0 * Cin * Cout, the first ranges with no
 * epilogue at all (scale = shift = residual = NULL, act = none: `out` then IS the raw accumulator), the last one with
 * acc_init = that buffer and the layer's epilogue.  The matrix op takes acc_init as its C operand, so every output
 * element is still ONE fma chain over (k ascending, c ascending): the result has the bits of the single launch. */
int sv_conv_fwd_acc(const float* in, int64_t V_in, int64_t in_ld, int Cin, const float* W, int K, int Cout, const int32_t* perm,
                    const int32_t* nbr_s, const uint32_t* submask, const int32_t* tile_order, int64_t V_out, int64_t Vpad,
                    const float* acc_init, int64_t acc_ld, const float* scale, const float* shift, const float* residual,
                    int64_t res_ld, int act, float slope, float* out, int64_t out_ld, sv_stream_t stream);
/* Kernel instance the calling thread's last sv_conv_fwd launched: "name|fast=F,ring=R,full=U" (fast = buffer-addressed
 * form; a tensor beyond its 2 GB extent, a misaligned plan or an odd channel count takes the guarded form).  Tests and the
 * bench's per-kernel table read it back instead of re-deriving the dispatch. */
const char* sv_conv_last_instance(void);
/* Dispatch thresholds of the calling thread's later sv_conv_fwd calls.  The instance lists were measured one launch at a
 * time; inside a multi-stream frame pipeline taller tiles win earlier (their launch tails are filled by the neighbour
 * frames' kernels), so the library default scales every "chosen from N workgroups" threshold by 0.3.  A caller that runs
 * ONE frame at a time (the reference's consumer: InferenceEngine.predict per frame, app/main.py:432-456) sets
 * want_scale = 1 - which also selects, for the thin 32 -> 32 layers, the kernel that keeps the layer's weights in LDS (one
 * 16-wave workgroup per CU: it needs whole CUs, which only a GPU that holds one frame has free).
 * tail_fraction = share of the plan tiles that chip-filling launches run as half-height tiles.
 * A negative value restores the library default (environment SV_CONV_WANT_SCALE / SV_CONV_TAIL).  Results never depend on
 * the instance (one fma chain per output element in every one of them). */
int sv_conv_set_dispatch(double want_scale, double tail_fraction);

/* Stand-alone BN(eval)/bias + residual + activation on feature rows, same arithmetic as the conv epilogue:
 *   out[v][c] = act( fmaf(in[v][c], scale[c], shift[c]) + residual[v][c] )
 * (ME.MinkowskiBatchNorm / MinkowskiReLU / MinkowskiLeakyReLU when not fused behind a conv, e.g.
 *  model/robotnet.py:47-50 output_layer, model/robotnet_segmentation.py:60) */
int sv_affine_act(const float* in, int64_t in_ld, int C, int64_t V, const float* scale, const float* shift,
                  const float* residual, int64_t res_ld, int act, float slope, float* out, int64_t out_ld,
                  sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A11  pre-voxelisation transforms on device rows
 *   (replace utils/preprocess.py:8-11 center_at_origin, :14-17 base_at_origin, :20-37 normalize_colors,
 *    :40-56 normalize_points; callers app/inference_engine.py:396-404,440-444,470-488)
 * ------------------------------------------------------------------------------------------- */
/* Column statistics of x[N][C], C <= 4: col_min[C], col_max[C] (exact), col_sum[C] (float64, deterministic order) and,
 * when max_row_norm is given, max over rows of the float32 norm of (row - sub) in numpy's order
 * sqrt(((x0-s0)^2 + (x1-s1)^2) + (x2-s2)^2).  workspace: sv_col_stats_workspace_bytes(N). */
size_t sv_col_stats_workspace_bytes(int64_t N);
int sv_col_stats(const float* x, int64_t ld, int64_t N, int C, const float* sub, void* workspace, size_t workspace_bytes,
                 float* col_min, float* col_max, double* col_sum, float* max_row_norm, sv_stream_t stream);
/* out[r][c] = (x[r][c] - sub[c]) / div[c] + add[c]; a null sub / div / add skips that operation (IEEE float32 operations
 * in this order: `points - offset` and `rgb / 255` round exactly as the reference's numpy expressions). */
int sv_center_scale(const float* x, int64_t ld, int64_t N, int C, const float* sub, const float* div, const float* add,
                    float* out, int64_t out_ld, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A6/A7  pooling, slice, argmax
 *   (replace ME.MinkowskiGlobalMaxPooling/AvgPooling model/robotnet.py:43, robotnet_encode.py:41;
 *    SparseTensor.slice app/inference_engine.py:417,551; utils/output.py:67-73)
 * ------------------------------------------------------------------------------------------- */
/* batch_start[b] = first row with batch index >= b, b = 0..B  (rows are sorted by batch first). */
int sv_batch_offsets(const uint64_t* keys, int64_t V, int B, int32_t* batch_start, sv_stream_t stream);
int sv_global_pool(const float* F, int64_t ld, int C, const int32_t* batch_start, int B, int mode, float* out,
                   sv_stream_t stream);
int sv_slice_rows(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, float* out,
                  sv_stream_t stream);
/* label[i] = first index of the row maximum of F[inverse[i]][0..C), conf[i] = sigmoid(max). */
int sv_slice_argmax(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, int64_t* label,
                    float* conf, sv_stream_t stream);
/* Key-point selection (utils/output.py:81-87 get_key_point_predictions): softmax over the C <= 32 classes of each of
 * the N rows of `logits`, then per class c: prob[c] = max over rows of softmax[:, c], idx[c] = the LOWEST row that attains
 * it (-1 and 0 when N = 0), selected[c] = prob[c] > conf_th.  workspace: 8 C bytes.  One pass over the logits, no host
 * round trip between softmax, max and threshold. */
int sv_key_point_predictions(const float* logits, int64_t ld, int C, int64_t N, float conf_th, void* workspace,
                             size_t workspace_bytes, float* prob, int64_t* idx, int32_t* selected, sv_stream_t stream);

/* The same selection for G clouds whose rows are consecutive segments of `logits` (the crops of G frames in one sparse
 * tensor): segment g = rows seg_start_host[g] .. seg_start_host[g + 1] (HOST array, G + 1 entries); outputs [G][C], idx
 * relative to the segment's first row.  workspace: 8 C G bytes.  Segment by segment the result of the single call. */
int sv_key_point_predictions_batched(const float* logits, int64_t ld, int C, const int64_t* seg_start_host, int G, float conf_th,
                                     void* workspace, size_t workspace_bytes, float* prob, int64_t* idx, int32_t* selected,
                                     sv_stream_t stream);
/* idx[j] = row of the j-th largest x[i * ld], i < N, j < k <= 64 (ties: the lower row first; -1 when N < k; NaN orders
 * above +inf as in torch.sort) - the `out[:, 1].sort(descending=True)[1][:8]` of utils/output.py:45-64 get_pred_center
 * without sorting N votes.  workspace: sv_topk_workspace_bytes(N, k). */
size_t sv_topk_workspace_bytes(int64_t N, int k);
int sv_topk_indices(const float* x, int64_t ld, int64_t N, int k, void* workspace, size_t workspace_bytes, int64_t* idx,
                    sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A9/A10/A12  dense solves, one wavefront per problem, float64
 *   (replace utils/transformation.py:178-222 get_rigid_transform_3D + :80-84 get_q_from_matrix,
 *    utils/calibration.py:69-95 compute_quaternions_weighted_average, utils/metrics.py:139-150 compute_ADD_np)
 * ------------------------------------------------------------------------------------------- */
/* ref,tgt: double[B][Kmax][3]; K: int32[B] points used per problem (>= 3).  R: double[B][9] row-major,
 * t: double[B][3], q: double[B][4] (w,x,y,z), sign as scipy's Rotation.from_matrix leaves it. */
int sv_kabsch_batched(const double* ref, const double* tgt, const int32_t* K, int Kmax, int B, double* R, double* t,
                      double* q_wxyz, sv_stream_t stream);
/* Q: double[B][Mmax][4] (w,x,y,z), w: double[B][Mmax], M: int32[B].  out: double[B][4] principal eigenvector of
 * sum w_i q_i q_i^T / sum w_i, normalised, sign such that the largest-magnitude component is positive. */
int sv_quat_avg_batched(const double* Q, const double* w, const int32_t* M, int Mmax, int B, double* out,
                        sv_stream_t stream);
/* ADD = mean_p || (R_gt p + t_gt) - (R_pr p + t_pr) ||, poses (x,y,z,qw,qx,qy,qz).  points double[B][Pmax][3]. */
int sv_add_metric_batched(const double* points, const int32_t* P, int Pmax, const double* gt_pose,
                          const double* pred_pose, int B, double* add_out, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N3  point-to-point ICP refinement (replaces utils/icp.py:13-83 = open3d registration_icp with
 *      TransformationEstimationPointToPoint; call sites app/inference_engine.py:358-362)
 *   src float32[S][3] (CAD model points), tgt float32[T][3] (end-effector crop), init_T double[16] row-major 4x4
 *   source->target (NULL = identity).  Correspondence = nearest target point within max_distance; update = Kabsch on
 *   the correspondences; stops when |d fitness| < rel_fitness and |d rmse| < rel_rmse between two evaluations or after
 *   max_iterations updates.  out_T double[16]; out_stats double[3] = {fitness, inlier rmse, updates applied}.
 *   The whole iteration runs on the stream without host read-backs.
 * ------------------------------------------------------------------------------------------- */
size_t sv_icp_workspace_bytes(int64_t S);
int sv_icp_point2point(const float* src, int64_t S, const float* tgt, int64_t T, const double* init_T,
                       double max_distance, int max_iterations, double rel_fitness, double rel_rmse, void* workspace,
                       size_t workspace_bytes, double* out_T, double* out_stats, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A8  PointNet++ sampling / grouping  (replace model/pointnet2_utils.py:65-86 farthest_point_sample,
 *      :89-109 query_ball_point, utils/data.py:13-34 numpy FPS)
 * ------------------------------------------------------------------------------------------- */
/* xyz float32[B][N][3]; start int64[B] first centroid (the reference draws it at random: pass it in);
 * out int64[B][S].  Distances are float32 ((dx*dx + dy*dy) + dz*dz, no fma), argmax = first maximum. */
int sv_fps(const float* xyz, int B, int N, int S, const int64_t* start, int64_t* out, sv_stream_t stream);
/* out int64[B][S][nsample]: the first nsample indices n (ascending) with ||xyz[n]-new_xyz[s]||^2 <= r^2,
 * padded with the first hit (r^2 = (float)(radius*radius), as torch compares a float32 tensor with the python
 * scalar radius**2); the distance uses the reference's expanded form (-2ab + a^2 + b^2) in float32. */
int sv_ball_query(const float* xyz, const float* new_xyz, int B, int N, int S, double radius, int nsample,
                  int64_t* out, sv_stream_t stream);
/* PointNetFeaturePropagation's interpolation (model/pointnet2_utils.py:298-305): out[b][n][:] = sum over the three
 * nearest xyz2 points of points2 rows weighted by 1 / (d + 1e-8), normalised; d in the reference's expanded float32
 * form, nearest first, ties to the lower index.  xyz1 float32[B][N][3], xyz2 [B][S][3] (S >= 3), points2 [B][S][C],
 * out [B][N][C]. */
int sv_three_nn_interpolate(const float* xyz1, const float* xyz2, const float* points2, int B, int N, int S, int C,
                            float* out, sv_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N4  largest single-linkage cluster of the end-effector points
 *   (replaces utils/output.py:13-28 ClusterUtil.get_largest_cluster = sklearn AgglomerativeClustering(
 *    linkage="single", distance_threshold=0.06) + most frequent label; call site app/inference_engine.py:422-433)
 *
 * Single linkage cut at `dist` = connected components of the graph "distance < dist" (strict, as sklearn merges while
 * the linkage distance is below the threshold).  Points: rows idx[i] (idx NULL = row i) of xyz, float32 or float64
 * (elem_bytes 4 / 8), row stride ld elements; distance in float64: sqrt((dx*dx + dy*dy) + dz*dz), no fma.
 *   root[i]  = the smallest i' in i's component (i, i' positions 0..n-1): a deterministic labelling
 *   best[0]  = root of the largest component (ties: the smallest root), best[1] = its size   (best may be NULL)
 * Lock-free union-find over all n(n-1)/2 pairs, tiled through LDS.  workspace: sv_cluster_workspace_bytes(n).
 * ------------------------------------------------------------------------------------------- */
size_t sv_cluster_workspace_bytes(int64_t n);
int sv_single_linkage_roots(const void* xyz, int elem_bytes, int64_t ld, const int32_t* idx, int64_t n, double dist,
                            void* workspace, size_t workspace_bytes, int32_t* root, int32_t* best, sv_stream_t stream);
/* out = the ascending positions i with v[i] == value (v int32 / int64 by elem_bytes; value_dev non-NULL: compare with
 * the int32 it points to on the device instead of `value`), count[0] = how many.  The np.where(...)[0] of
 * utils/output.py:24 and app/inference_engine.py:420 without a host round trip. */
int sv_select_equal(const void* v, int elem_bytes, int64_t n, int64_t value, const int32_t* value_dev, int64_t* out,
                    int64_t* count, sv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SV_HIP_H */
