/*
 * ocrf_hip.h — C ABI of libocrf_hip.so: the MI355X (gfx950) implementation of OcRFDet's
 * render + BEV-pool + HOA hot path.  Plain pointers and sizes only; every pointer is a DEVICE
 * pointer unless its comment says "host".  Every function returns 0 on success or a hipError_t
 * value (as int); nothing is allocated, freed or synchronised inside a call (all of them are
 * hipGraph-capturable), the caller owns every buffer, and `stream` is a hipStream_t (NULL =
 * the legacy default stream, which is what the reference launches on).
 *
 * File:line citations are into the reference tree (Mingqj/OcRFDet), i.e. the interface each
 * entry point replaces.
 */
#ifndef OCRF_HIP_H
#define OCRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ocrf_stream_t; /* hipStream_t */

/* Library / build identification: returns e.g. "ocrf_hip 0.1 gfx950". Host pointer, static. */
const char *ocrf_version(void);

/* ------------------------------------------------------------------------------------------
 * BEVPoolv2 voxel pooling
 * ------------------------------------------------------------------------------------------ */

/*
 * Exact signature of the reference's C-level launcher
 *   void bev_pool_v2(int c, int n_intervals, const float* depth, const float* feat,
 *                    const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
 *                    const int* interval_starts, const int* interval_lengths, float* out)
 * (mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:7-9, defined src/bev_pool_cuda.cu:123-131).
 * Same contract: `out` is (B,Z,Y,X,C) and must be pre-zeroed by the caller; for interval k,
 *   out[ranks_bev[starts[k]]*c + ch] = sum_i depth[ranks_depth[starts[k]+i]] * feat[ranks_feat[starts[k]+i]*c + ch]
 * (assignment, accumulated in list order); launches on the legacy default stream; any interval
 * layout is accepted (one lane group per interval).  No error is reported (the reference
 * reports none) — use ocrf_bev_pool_v2 for an error code, a stream and the tiled kernel.
 */
void bev_pool_v2(int c, int n_intervals, const float *depth, const float *feat,
                 const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                 const int *interval_starts, const int *interval_lengths, float *out);

/*
 * Exact signature of the reference's backward launcher (src/bev_pool.cpp:11-14,
 * src/bev_pool_cuda.cu:133-140).  Intervals are runs of equal ranks_feat (the Python wrapper
 * re-sorts, bev_pool.py:47-57).  depth_grad/feat_grad must be pre-zeroed.
 *   feat_grad[ranks_feat[s]*c + ch] = sum_i out_grad[ranks_bev[s+i]*c + ch] * depth[ranks_depth[s+i]]
 *   depth_grad[ranks_depth[p]]      = sum_ch out_grad[ranks_bev[p]*c + ch] * feat[ranks_feat[p]*c + ch]
 */
void bev_pool_v2_grad(int c, int n_intervals, const float *out_grad, const float *depth,
                      const float *feat, const int *ranks_depth, const int *ranks_feat,
                      const int *ranks_bev, const int *interval_starts,
                      const int *interval_lengths, float *depth_grad, float *feat_grad);

/*
 * Tiled forward (one output-stationary pass: csrc/bev_pool.hip).  Same arithmetic contract as bev_pool_v2 —
 * `out` is (B,Z,Y,X,C) = (n_voxels, C) rows — with these additions / differences:
 *   n_points        length of the three rank vectors (the reference passes it implicitly as
 *                   tensor sizes, bev_pool.cpp:30-57);
 *   n_voxels        rows of `out` (B*Z*Y*X, < 2^30); EVERY row is written (empty voxels as 0), so `out`
 *                   needs no pre-zeroing (pre-zeroed, as the reference's caller does, is fine too);
 *   workspace       scratch of at least ocrf_bev_pool_v2_workspace_bytes(c, n_points, n_voxels) bytes,
 *                   16-byte aligned; contents are don't-care on entry and exit;
 *   stream          hipStream_t.
 * ANY interval layout is accepted, like the reference's kernel: unsorted, overlapping, empty intervals, gaps
 * (points outside every interval are ignored).  Two intervals that name the same voxel race in the reference
 * (two threads assign the same row); here one of them wins as well.  Channel counts outside C % 4 == 0,
 * 32 <= C <= 256 take the reference's own thread mapping (then `out` must be pre-zeroed).
 * Results are bitwise reproducible run to run (no float atomics); a voxel whose points lie inside one 32-point
 * sub-chunk of its tile's point list is summed in exactly the reference's order.
 */
int ocrf_bev_pool_v2(int c, int n_intervals, int n_points, long n_voxels, const float *depth, const float *feat,
                     const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                     const int *interval_starts, const int *interval_lengths, float *out,
                     void *workspace, size_t workspace_bytes, ocrf_stream_t stream);

size_t ocrf_bev_pool_v2_workspace_bytes(int c, int n_points, long n_voxels);

/*
 * Fused forward: the same pass writing the layout the reference reaches with its passes AFTER the op, directly:
 *   layout 0: out[b][c][z][y][x]      == bev_pool_v2(...) of the reference, i.e. the op's
 *                                        permute(0,4,1,2,3).contiguous() (bev_pool.py:91);
 *   layout 1: out[b][z*C + c][y][x]   == torch.cat(bev_feat.unbind(dim=2), 1), what
 *                                        voxel_pooling_v2 / fast_sampling return
 *                                        (view_transformer.py:194, view_transformer_ocrf.py:781).
 * Every element of `out` is written exactly once (64 voxels x C channels per workgroup, 256-byte runs per
 * channel).  (B,Z,Y,X) is the reference's bev_feat_shape without C (B*Z*Y*X < 2^30).  Needs C % 4 == 0,
 * 32 <= C <= 256; ranks_bev values outside [0, B*Z*Y*X) are ignored.
 * workspace >= ocrf_bev_pool_v2_nchw_workspace_bytes(c, n_intervals, n_points, B, Z, Y, X).
 */
int ocrf_bev_pool_v2_nchw(int c, int n_intervals, int n_points, const float *depth,
                          const float *feat, const int *ranks_depth, const int *ranks_feat,
                          const int *ranks_bev, const int *interval_starts,
                          const int *interval_lengths, float *out, int B, int Z, int Y, int X,
                          int layout, void *workspace, size_t workspace_bytes,
                          ocrf_stream_t stream);

/*
 * Plans for rank vectors that stay the same across calls (static calibration: the reference's
 * `accelerate=True`, pre_compute at view_transformer.py:257-262 / view_transformer_ocrf.py:854-866).
 * ocrf_bev_pool_plan_build runs the rank-only part of the pooling once — the dense voxel table, the list of
 * work units (tiles and slices of heavy tiles), their split over the XCDs — and keeps it in `plan` (device
 * memory, caller-owned, >= ocrf_bev_pool_plan_bytes(c, n_points, B, Z, Y, X) bytes, 16-byte aligned);
 * ocrf_bev_pool_v2_nchw_planned is ocrf_bev_pool_v2_nchw for exactly those rank vectors (same c, n_points,
 * grid) as ONE launch: same results bit for bit.  ranks_bev / interval_* are not needed again after the build.
 * The plan holds the arrival counters of the cut tiles (left at zero by every call), so a plan serves one
 * stream at a time.  workspace >= ocrf_bev_pool_planned_workspace_bytes(c, n_points).
 */
size_t ocrf_bev_pool_plan_bytes(int c, int n_points, int B, int Z, int Y, int X);
int ocrf_bev_pool_plan_build(int c, int n_intervals, int n_points, const int *ranks_bev,
                             const int *interval_starts, const int *interval_lengths, int B, int Z, int Y, int X,
                             void *plan, size_t plan_bytes, ocrf_stream_t stream);
size_t ocrf_bev_pool_planned_workspace_bytes(int c, int n_points);
int ocrf_bev_pool_v2_nchw_planned(int c, int n_points, const float *depth, const float *feat,
                                  const int *ranks_depth, const int *ranks_feat, void *plan, float *out,
                                  int B, int Z, int Y, int X, int layout, void *workspace,
                                  size_t workspace_bytes, size_t depth_bytes, size_t feat_bytes, ocrf_stream_t stream);
/* depth_bytes / feat_bytes: the sizes of the two operand tensors in bytes (0: not stated).  The kernels address them with
 * 32-bit byte offsets (the reference indexes with int, bev_pool_cuda.cu:39-47): an operand or an output of 2 GiB or more is
 * refused (hipErrorInvalidValue), and a stated feat_bytes bounds the kernel's feature loads (a rank beyond it reads 0). */

/* Same, for rank vectors whose lengths were produced on the device (ocrf_lss_prepare /
 * ocrf_ht_prepare): the vectors are passed at their capacities and `counts` (device, int32) holds
 * [n_points, n_intervals]; the launches are sized for the capacities and workgroups past the real
 * end retire at once, so nothing between index preparation and pooling reads the device.
 * Workspace: ocrf_bev_pool_v2_nchw_workspace_bytes(c, cap_intervals, cap_points, B, Z, Y, X). */
int ocrf_bev_pool_v2_nchw_dyn(int c, int cap_intervals, int cap_points, const int *counts,
                              const float *depth, const float *feat, const int *ranks_depth,
                              const int *ranks_feat, const int *ranks_bev, const int *interval_starts,
                              const int *interval_lengths, float *out, int B, int Z, int Y, int X,
                              int layout, void *workspace, size_t workspace_bytes, ocrf_stream_t stream);
size_t ocrf_bev_pool_v2_nchw_workspace_bytes(int c, int n_intervals, int n_points, int B, int Z, int Y, int X);

/*
 * Writes *flag (device int) = 0 if interval_starts/lengths are the ascending, non-overlapping cover the
 * reference's producers build (view_transformer.py:240-255) for n_points points, else a non-zero bit mask
 * (1: not ascending / overlapping, 2: interval exceeds n_points, 4: negative start or length).
 */
int ocrf_bev_pool_v2_check_intervals(int n_intervals, int n_points, const int *interval_starts,
                                     const int *interval_lengths, int *flag, ocrf_stream_t stream);

/* Stream-taking, error-returning form of bev_pool_v2_grad (same arithmetic; deterministic). */
int ocrf_bev_pool_v2_grad(int c, int n_intervals, const float *out_grad, const float *depth,
                          const float *feat, const int *ranks_depth, const int *ranks_feat,
                          const int *ranks_bev, const int *interval_starts,
                          const int *interval_lengths, float *depth_grad, float *feat_grad,
                          ocrf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gaussian rasteriser forward (RGB + depth + transmittance), batched over views
 * ------------------------------------------------------------------------------------------
 * Replaces `rasterize_gaussians` of the w-depth `diff_gaussian_rasterization` extension,
 *   std::tuple<int, Tensor color, Tensor radii, Tensor geomBuffer, Tensor binningBuffer,
 *              Tensor imgBuffer[, Tensor depth]>
 *   RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations,
 *                          scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
 *                          tan_fovy, image_height, image_width, sh, degree, campos, prefiltered)
 * (mmdet3d/models/necks/MVSGaussian/lib/submodules/diff-gaussian-rasterization/
 *  rasterize_points.cu:35-115), for the argument combination OcRFDet uses
 * (gaussian_renderer/__init__.py:62-70): colours precomputed (no SH evaluation), covariance from
 * (scales, rotations) or cov3D_precomp.  One call renders n_views cameras over the same P
 * Gaussians (n_views = 1 is the reference's call).
 *
 *   means3D (P,3)  colors (P,3)  opacities (P)  scales (P,3)  rotations (P,4) [r,x,y,z; not
 *   normalised, forward.cu:127]  cov3D_precomp (P,6) or NULL (then scales/rotations are used)
 *   cameras (n_views, 36) floats: viewmatrix[16] | projmatrix[16] | tanfovx | tanfovy |
 *           focal_x = W/(2 tanfovx) | focal_y = H/(2 tanfovy); both matrices are the TRANSPOSED
 *           (row-vector) 4x4 the reference passes (auxiliary.h:58-77)
 *   bg (3)         depth_mode 0 = median depth (w-depth default, 15.0 where T never crosses 0.5),
 *                             1 = mean depth (the fork's commented-out alternative)
 * outputs (caller-allocated, fully written):
 *   out_color (n_views,3,H,W)  out_depth (n_views,H,W)  out_final_T (n_views,H,W) [= 1 - accumulated
 *   opacity]  out_n_contrib (n_views,H,W) or NULL (inference: only the backward reads it; NULL drops its
 *   tracking from the blend)  radii (n_views,P)  tiles_touched (n_views,P) or NULL
 *   status (device int, may be NULL; written by the call): informational.  Bit 1 (value 2) is set when
 *   some tile met more than ~1 000 Gaussians inside ONE 0.2 %-wide depth bucket, more than the in-LDS
 *   sort holds; such a tile is then blended by an exact but slower streaming selection over that
 *   bucket.  Results are exact either way; bit 0 is never set (reserved).
 * P == 0 zero-fills the outputs like the reference (rasterize_points.cu:68-69).
 * No host synchronisation happens (the reference reads num_rendered back, rasterizer_impl.cu:281).
 */
int ocrf_rasterize_forward(int P, int n_views, int H, int W, const float *means3D,
                           const float *colors, const float *opacities, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *cameras, const float *bg, int depth_mode, float *out_color,
                           float *out_depth, float *out_final_T, uint32_t *out_n_contrib, int *radii,
                           uint32_t *tiles_touched, int *status, void *workspace,
                           size_t workspace_bytes, ocrf_stream_t stream);

/*
 * The same forward over n_sets Gaussian sets of P Gaussians each in ONE set of launches: the arrays
 * are (n_sets, P, .), `cameras` holds n_sets * views_per_set rows and view v renders set
 * v / views_per_set (the per-sample loop of view_transformer_ocrf.py:1090-1153 — every sample has its
 * own Gaussian parameters and one camera — or the frames of a multi-frame batch).  Outputs are indexed
 * by view as above; workspace as for n_sets * views_per_set views.
 */
int ocrf_rasterize_forward_sets(int P, int n_sets, int views_per_set, int H, int W, const float *means3D,
                                const float *colors, const float *opacities, const float *scales,
                                float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                const float *cameras, const float *bg, int depth_mode, float *out_color,
                                float *out_depth, float *out_final_T, uint32_t *out_n_contrib, int *radii,
                                uint32_t *tiles_touched, int *status, void *workspace,
                                size_t workspace_bytes, ocrf_stream_t stream);

size_t ocrf_rasterize_workspace_bytes(int P, int n_views);

/*
 * Static render plans (csrc/raster_plan.hip).  In OcRFDet the Gaussian means are the fixed voxel grid
 * (view_transformer_ocrf.py:651-673,690-692) and the cameras are fixed per calibration; what depends on
 * (mean, camera) alone — the near-plane cull and view-space depth (auxiliary.h:139-164), the projected centre and
 * the Jacobian of computeCov2D (forward.cu:83-98,196-199), hence the whole (depth bits, id) blend order the
 * reference obtains by sorting every step (rasterizer_impl.cu:226-267) — is computed ONCE:
 *
 *   1. ocrf_raster_plan_count     (sizing, optional) how many (Gaussian, view) records a plan for these cameras keeps:
 *      g_mask (device, P words; bit v = view v keeps the Gaussian) and their number in *total (device).  The caller
 *      reads the total once and chooses a record CAPACITY (with headroom if the plan will be rebuilt for other poses).
 *   2. ocrf_raster_plan_build     writes the plan (device, >= ocrf_raster_plan_bytes(P, n_views, capacity) bytes,
 *      256-byte aligned): header, cameras; per view the kept records in blend order (id, depth bits, pixel centre);
 *      per Gaussian the views that keep it (never visible = behind the near plane, or outside the frame for every
 *      world-space extent <= extent_bound) and, per (Gaussian, view), the record's place in that order with the rows
 *      of J W (forward.cu:83-98).  NO host read and kernels only (hipGraph-capturable): classify all views ->
 *      exclusive scan -> records + one 32-bit sort key per record (view << 27 | depth bits - 0x3E000000: view-space
 *      depths in [0.125, 8191) m) -> ONE stable radix sort -> gather.  It may be called again on the same buffers
 *      with other cameras — the reference recomputes the render cameras from the dataloader's c2w for every sample
 *      (view_transformer_ocrf.py:1140-1152) — a rebuild per sample is ~ 20 launches.  A plan whose records exceed
 *      `capacity`, a depth outside the key range or a scan that gave up leave the plan marked unusable ON THE DEVICE
 *      (every render of it raises status bit 8).  workspace >= ocrf_raster_plan_build_workspace_bytes(P, n_views,
 *      capacity).
 *
 * extent_bound bounds scale_modifier * max_k |s_k| * |R(q)|_2 (|R(q)|_2 = |1 - |q|^2| + |q|^2: 1 for a normalised
 * quaternion) of every Gaussian the plan will be rendered with; a plan stays valid for ANY parameters within it.
 *
 *   3. ocrf_rasterize_planned     one render = the front end + the blend of the sorted lists.  Front end, when radii
 *      are an output (radii != NULL with guard 0, or guard bit 1): one thread per Gaussian builds its 3D covariance once
 *      and writes conic / tile rect of EVERY rendered view's record (record order: coalesced; the blend walks a view's
 *      depth-ordered list through the plan's static list -> record map).  Else (round 5): only the HEAD of every rendered
 *      view's list is prepared, in list order (a tile pair stops scanning once its pixels are saturated — at cfg2 after
 *      ~500 of 120 000 entries); a tile pair that scans further extends the arrays itself, and the head's length
 *      follows what the previous call needed (kept in `workspace`, no host read).  The extent check of all Gaussians
 *      then runs behind the blend (a status bit) — or in front of it with the device guard.
 *      n_items views are rendered; item z renders plan view item_view[z] (device ints; NULL = z, then n_items /
 *      n_sets must equal the plan's view count) with Gaussian set z / (n_items / n_sets) of the (n_sets, P, .)
 *      parameter arrays; the items of one set name distinct views.  Outputs as
 *      ocrf_rasterize_forward, indexed by item; colour / depth / final_T are bit-identical to it.  radii
 *      (n_items, P) or NULL.  `capacity`: the plan's record capacity (as given to the build).
 *      workspace >= ocrf_rasterize_planned_workspace_bytes(capacity, n_sets).
 *      views_disjoint = 1: the caller states that every plan view is rendered by at most ONE set of this call (frames
 *      of a sample sharing a plan, each with its own parameter set): all sets then share one copy of the per-call
 *      record arrays (workspace >= ocrf_rasterize_planned_workspace_bytes(capacity, 1)), every line of which is
 *      written once; a view named by two sets raises status bit 3 (value 8).  0: a copy per set.
 *      call_cameras (device, n_plan_views x 36 floats, or NULL): the cameras THIS call means to render with.  The
 *      head kernel compares them with the plan's, bit for bit, on the device; a difference raises status bit 4
 *      (value 16) and — with guard = 1 — hands the call to the armed per-call pipeline, which then renders with
 *      call_cameras: a stale plan can not silently render another pose.
 *      status (device int, NOT written unless something is wrong: zero it once): bit 2 (value 4) = some Gaussian
 *      exceeded extent_bound this call, bit 3 (value 8) = bad item_view / unusable plan.
 *      guard = 0: with bit 2 set the outputs of that call are not valid (the caller re-renders with
 *      ocrf_rasterize_forward or rebuilds the plan with a larger bound).
 *      guard = 1 (+ 2: the radii are an output too; without it the radii buffer is the armed chain's scratch only):
 *      the per-call pipeline of ocrf_rasterize_forward is enqueued behind the planned one, armed by the
 *      extent check: all of its kernels retire at once when the bound holds, and render the call instead when it
 *      does not — exact results either way, no host involvement (hipGraph-capturable), at the price of four
 *      near-empty launches (no memsets: the flag is raised by the extent check / head kernel and lowered by the armed blend, the
 *      chain's histograms are cleared by the planned blend when the flag is up, radii zeros come from the update
 *      kernel).  Needs means3D, radii and chain_workspace >= ocrf_rasterize_workspace_bytes(P, n_items); `workspace`
 *      must be zero-filled when it is allocated and belong to this plan alone (it carries the flag between calls; a
 *      stale non-zero flag costs one slow, still exact, call).
 *      blend_workgroups: size of the blend's persistent grid (its workgroups draw tile pairs from a ticket queue);
 *      0 = as many as the device holds at once (fastest alone).  The blend is VALU-bound: two workgroups per CU keep
 *      most of its speed and leave the other wave slots to kernels of other streams (the hot path renders beside its
 *      latency-bound poolings with 2 x CUs: -13 % step time at cfg2).
 *      yield_if (device int or NULL): a scheduling hint.  With it the blend is launched on every slot the device has
 *      and the workgroups beyond `blend_workgroups` leave at once while *yield_if != 0 — the caller raises the word
 *      (ocrf_stream_write_value32) while another stream's chain should keep those wave slots and lowers it when that
 *      chain is done, so that later renders of the same step take the whole chip.  Any value renders the same image.
 *      phase: 0 = both launches; 1 = only the update, 2 = only the blend of a call whose update already ran (guard 0
 *      only) — the two may then sit on different streams (the update is memory-bound, the blend VALU-bound: frame
 *      n + 1's update runs under frame n's blend), ordered by the caller's events.
 * Forward only (no n_contrib): training renders through ocrf_rasterize_forward / _backward.
 */
size_t ocrf_raster_plan_count_workspace_bytes(int P);
int ocrf_raster_plan_count(int P, int n_views, int H, int W, const float *means3D, const float *cameras,
                           float extent_bound, unsigned *g_mask, int *total, void *workspace,
                           size_t workspace_bytes, ocrf_stream_t stream);
size_t ocrf_raster_plan_build_workspace_bytes(int P, int n_views, long capacity);
size_t ocrf_raster_plan_bytes(int P, int n_views, long capacity);
int ocrf_raster_plan_build(int P, int n_views, int H, int W, const float *means3D, const float *cameras,
                           float extent_bound, long capacity, void *workspace, size_t workspace_bytes, void *plan,
                           size_t plan_bytes, ocrf_stream_t stream);
size_t ocrf_rasterize_planned_workspace_bytes(long capacity, int n_sets);
int ocrf_rasterize_planned(const void *plan, size_t plan_bytes, int P, int n_plan_views, long capacity,
                           int H, int W, int n_sets, int n_items, const int *item_view,
                           const float *colors, const float *opacities, const float *scales, float scale_modifier,
                           const float *rotations, const float *bg, int depth_mode, float *out_color,
                           float *out_depth, float *out_final_T, int *radii, int *status, void *workspace,
                           size_t workspace_bytes, int guard, const float *means3D, void *chain_workspace,
                           size_t chain_workspace_bytes, int blend_workgroups, const int *yield_if, int phase,
                           const float *call_cameras, int views_disjoint, const void *bins, size_t bins_bytes,
                           int bin_w, int bin_h, long cand_capacity, int *hint, ocrf_stream_t stream);
/* hint (or NULL): one int in memory BOTH the host and the device can address (pinned host memory: hipHostMalloc), zero at
 * first.  The last workgroup of a render writes "some rendered view is deep" into it; the next call reads it ON THE HOST,
 * without a copy or a wait, to choose between the two builds of the blend — the plain one, and the one that compacts the
 * candidates of deep views per call (see ocrf_rasterize_planned_bins_workspace_bytes).  A stale value costs time, never
 * correctness; a replayed step (ocrf_step_*) re-reads it at every replay. */
/*
 * Candidate lists of a built plan — the static part of what the reference builds per call as its per-tile lists
 * (cuda_rasterizer/rasterizer_impl.cu:70-138 duplicateWithKeys + identifyTileRanges, consumed forward.cu:261-374): per
 * (plan view, bin of bin_w x bin_h tile PAIRS, i.e. bin_w x 2 bin_h tiles of 16 x 16 px) the positions, ascending (= blend
 * order), of the view's records whose tile rect for ANY parameters within the plan's extent bound reaches the bin.  With
 * them (bins != NULL in ocrf_rasterize_planned; bin_w / bin_h / cand_capacity as given to the build) a tile pair tests the
 * rects of its bin's candidates instead of the view's whole list: its cost is O(own records).  Same images bit for bit.
 *   ocrf_raster_plan_bins_build: `plan` built on this stream before; no host read, kernels only.  cand_capacity = 0 is a
 *   sizing pass (only *total_out, a device int, is written); lists that do not fit the capacity leave `bins` unusable — it
 *   is then ignored by renders (whole lists are walked), never wrong.  workspace >=
 *   ocrf_raster_plan_bins_workspace_bytes(...), bins >= ocrf_raster_plan_bins_bytes(...).
 * A planned render runs in two passes of the blend: the first renders every tile pair as far as the PREPARED head of its
 * view's list reaches (the head follows what the last call needed, up to the whole list) and hands the tile pairs that
 * need more to the second; between them the records behind the heads are prepared ONCE (inside the extent-check launch),
 * only if a tile pair asked.  A tile pair never prepares records itself.
 */
size_t ocrf_raster_plan_bins_bytes(int n_views, int H, int W, int bin_w, int bin_h, long cand_capacity);
/* per-call scratch of ocrf_rasterize_planned when it is given candidate lists: as ocrf_rasterize_planned_workspace_bytes
 * plus room for the per-call compaction of "deep" views — views whose last render walked 8 192 list entries or more
 * (pixels that do not saturate early: an object-centric opacity field).  For those the head of the list is the whole
 * list and, per segment of 1 024 candidates of a bin, the candidates whose rect of THIS call reaches the bin are moved to
 * the front with the mask of the bin's tiles they cover (the per-call half of duplicateWithKeys,
 * rasterizer_impl.cu:70-109), so that a tile pair reads only records that can concern it.  A workspace of the smaller
 * size is accepted when bins == NULL. */
size_t ocrf_rasterize_planned_bins_workspace_bytes(long capacity, int n_sets, int n_views, int H, int W, int bin_w,
                                                   int bin_h, long cand_capacity);
size_t ocrf_raster_plan_bins_workspace_bytes(int P, int n_views, int H, int W, int bin_w, int bin_h, long capacity);
int ocrf_raster_plan_bins_build(const void *plan, size_t plan_bytes, int P, int n_views, long capacity, int H, int W,
                                float extent_bound, int bin_w, int bin_h, long cand_capacity, void *bins,
                                size_t bins_bytes, int *total_out, void *workspace, size_t workspace_bytes,
                                ocrf_stream_t stream);

/*
 * bev_pool_v2 forward as per-tile MFMA panels (csrc/bev_pool_mfma.hip): out[64 voxels x C] = W[64 x R] . F[R x C]
 * with R the tile's UNIQUE feature rows, W[v][r] = the summed depth weights of the tile's points with voxel slot v
 * and row slot r, on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: an exact fmaf chain, bitwise reproducible).
 * Same result as ocrf_bev_pool_v2_nchw up to the summation order (reference semantics: bev_pool_cuda.cu:39-47).
 * The rank-only part is a plan the caller builds once (ocrfdet_amd/bevpool.MfmaPoolPlan shows how), all device ints:
 *   units (n_units x 4)      {tile, first panel, end panel, slice | n_slices << 16}: a tile's panels in groups; tiles
 *                            without points have one unit with no panel (they are written as zeros); every tile of the
 *                            (B*Z) planes x ceil(Y / 8) x ceil(X / 8) grid appears (tile = plane * tpp + ty * tx_count + tx)
 *   unit_slab (n_units)      first slab of the unit's tile (tiles of several units), else anything
 *   panel_rows (n_panels x R_p), panel_nrows (n_panels)   feature rows of a panel (R_p = ocrf_bev_pool_mfma_panel_rows())
 *   panel_cell_off (n_panels + 1), cells (n_cells x 4): {v | r << 8 | points << 16, depth ranks of the (up to three)
 *                            points in summation order}; a cell of more points: {v | r << 8 | 0xFFFF << 16, first
 *                            index into rd_sorted, points, -}; rd_sorted: depth ranks of all points in cell order
 *                            a unit holds at most ocrf_bev_pool_mfma_max_unit_panels() panels
 *   arrive (n_tiles ints, zero before the first call; left zero), slabs (>= ocrf_bev_pool_mfma_slab_bytes(c, slices))
 * C in {64, 80, 96, 128}; layouts as ocrf_bev_pool_v2_nchw (+ 2: rows (n_vox, C)).
 */
int ocrf_diag_pool_mfma_stamps(unsigned long long *buf); /* diagnostic: per-unit phase cycles of the next C = 80 calls */
int ocrf_bev_pool_mfma_panel_rows(void);
int ocrf_bev_pool_mfma_tile_side(void); /* tiles are side x side voxel blocks of a (b, z) plane, row-major; slot v = side * dy + dx */
size_t ocrf_bev_pool_mfma_slab_bytes(int c, int n_slab_slices);
int ocrf_bev_pool_mfma_max_unit_panels(void);
int ocrf_bev_pool_v2_nchw_mfma(int c, int n_units, const int *units, const int *unit_slab, const int *panel_rows,
                               const int *panel_nrows, const int *panel_cell_off, const int *cells,
                               const int *rd_sorted, const float *depth, const float *feat,
                               float *out, int B, int Z, int Y, int X, int layout, int *arrive, void *slabs,
                               ocrf_stream_t stream);

/*
 * The same plan as a latency kernel (csrc/bev_pool_panel.hip): the cells' depth weights are summed by a pre-pass
 * (ocrf_bev_pool_cell_weights: one thread per cell, up to two cell lists per launch — the LSS and the height-sampling
 * plan of a step read the same depth tensor; either list may be empty); in the pooling kernel a lane is a voxel slot
 * (a wave a quarter of the channels) and walks its cells out of LDS: acc = fma(F[r], w(v, r), acc) with rows ascending,
 * panel after panel — the k order of the MFMA form, bit-identical to it on finite inputs; bitwise reproducible.
 * Extra plan arrays (device): the cells of a panel are stored VOXEL-major (v ascending, then r —
 * ocrf_bev_pool_v2_nchw_mfma does not depend on the order inside a panel),
 *   panel_voff (n_panels x 64)   first cell of voxel slot v, relative to the panel's first cell
 *   cell_code (n_cells u16)      row slot | voxel slot << 8;     cw (n_cells floats)  the pre-pass's output
 */
int ocrf_diag_pool_panel_stamps(unsigned long long *buf); /* diagnostic: per-unit phase cycles of the next C = 80 calls */
int ocrf_bev_pool_cell_weights(int n_cells0, const int *cells0, const int *rd_sorted0, float *cw0, int n_cells1,
                               const int *cells1, const int *rd_sorted1, float *cw1, const float *depth,
                               ocrf_stream_t stream);
int ocrf_bev_pool_v2_nchw_panel(int c, int n_units, const int *units, const int *unit_slab, const int *panel_rows,
                                const int *panel_nrows, const int *panel_cell_off, const int *panel_voff,
                                const unsigned short *cell_code, const float *cw, const float *feat, float *out, int B,
                                int Z, int Y, int X, int layout, int *arrive, void *slabs, size_t feat_bytes,
                                ocrf_stream_t stream); /* feat_bytes: as for ocrf_bev_pool_v2_nchw_planned */

/*
 * Backward of the colour output of ocrf_rasterize_forward (the w-depth fork has no depth backward,
 * diff-gaussian-rasterization-w-depth/README.md:13).  Replaces RasterizeGaussiansBackwardCUDA
 * (rasterize_points.cu:117-196 -> rasterizer_impl.cu:338-434 -> cuda_rasterizer/backward.cu) for
 * colours precomputed and covariance from (scales, rotations); gradients are summed over the views.
 *   fwd_color / fwd_final_T / fwd_n_contrib: the forward's outputs for the same inputs
 *   dL_dcolor (n_views,3,H,W)
 * outputs (fully written): dL_dmeans3D (P,3)  dL_dcolors (P,3)  dL_dopacity (P)  dL_dscales (P,3)
 *   dL_drotations (P,4) [w.r.t. the un-normalised quaternion, like the reference]
 *   dL_dmeans2D (n_views,P,3) or NULL [NDC-space screen gradient, z = 0; the reference returns it
 *   for densification statistics]
 * Float atomics are used per (tile, Gaussian): results vary in the last bits run to run, like the
 * reference's (backward.cu:509-541).  Reference quirk kept: a clamped t.x / t.y is treated as a
 * constant (x_grad_mul / y_grad_mul, backward.cu:174-175).
 */
int ocrf_rasterize_backward(int P, int n_views, int H, int W, const float *means3D, const float *colors,
                            const float *opacities, const float *scales, float scale_modifier,
                            const float *rotations, const float *cameras, const float *bg,
                            const float *fwd_color, const float *fwd_final_T, const uint32_t *fwd_n_contrib,
                            const float *dL_dcolor, float *dL_dmeans3D, float *dL_dcolors,
                            float *dL_dopacity, float *dL_dscales, float *dL_drotations,
                            float *dL_dmeans2D, void *workspace, size_t workspace_bytes,
                            ocrf_stream_t stream);
/* The same with the 3-D covariance handed over (cov3D_precomp (P,6): xx xy xz yy yz zz, as the forward takes it):
 * the reference stops at dL/dcov3D then (backward.cu:346-396 without computeCov3D's backward, :598-613):
 * dL_dcov3D (P,6), fully written, off-diagonal entries carry both symmetric halves (backward.cu:251-260). */
int ocrf_rasterize_backward_cov3d(int P, int n_views, int H, int W, const float *means3D, const float *colors,
                                  const float *opacities, const float *cov3D_precomp, const float *cameras,
                                  const float *bg, const float *fwd_color, const float *fwd_final_T,
                                  const uint32_t *fwd_n_contrib, const float *dL_dcolor, float *dL_dmeans3D,
                                  float *dL_dcolors, float *dL_dopacity, float *dL_dcov3D, float *dL_dmeans2D,
                                  void *workspace, size_t workspace_bytes, ocrf_stream_t stream);
size_t ocrf_rasterize_backward_workspace_bytes(int P, int n_views);

/*
 * Spherical-harmonics colours (the `shs` argument of the reference's rasteriser; computeColorFromSH,
 * cuda_rasterizer/forward.cu:20-71, used by preprocessCUDA when colors_precomp is NULL, :240-247): for one camera centre
 * `campos` (3 floats on the device), colors[g] = max(0.5 + sum_{i < (deg+1)^2} basis_i(dir_g) * shs[g][i], 0) with dir_g the
 * unit vector from campos to means3D[g]; clamped (P,3) bytes record which channels were clamped (the backward zeroes their
 * gradient, backward.cu:33-36).  shs (P, max_coeffs, 3), deg in 0..3, max_coeffs >= (deg+1)^2.  The (P,3) colours feed
 * ocrf_rasterize_forward as its `colors` for that view.
 * Backward (backward.cu:20-140): dL_dshs (P, max_coeffs, 3) is fully written (zero above the active degree); the
 * direction's dependence on the mean is ADDED to dL_dmeans3D (P,3) (which already holds the rasteriser's part).
 */
int ocrf_sh_to_rgb(int P, int deg, int max_coeffs, const float *means3D, const float *campos, const float *shs,
                   float *colors, unsigned char *clamped, ocrf_stream_t stream);
int ocrf_sh_to_rgb_backward(int P, int deg, int max_coeffs, const float *means3D, const float *campos,
                            const float *shs, const unsigned char *clamped, const float *dL_dcolors,
                            float *dL_dmeans3D, float *dL_dshs, ocrf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Index preparation of the two poolings on the device (csrc/index_prep.hip).
 *
 * ocrf_lss_prepare replaces get_lidar_coor + voxel_pooling_prepare_v2
 * (mmdet3d/models/necks/view_transformer.py:108-147,197-255) for one batch of B samples x N cameras:
 *   frustum   (D,H,W,3) device   the create_frustum template (:77-106), (u, v, depth) per cell
 *   cams      (B*N,33)  device   per camera-frame, row-major 3x3s:
 *                                inv(post_rots) 9 | rots . inv(cam2imgs) 9 | post_trans 3 | trans 3 | bda 9
 *                                (the tiny per-camera algebra stays with the caller: the same torch
 *                                calls the reference makes at :128-146)
 *   grid_lower_host / grid_interval_host   3 floats each in HOST memory (create_grid_infos, :59-75)
 *   gx, gy, gz                    grid size (X, Y, Z)
 * outputs (device, int32; capacities: B*N*D*H*W for the three rank vectors, min(that, B*Z*Y*X) for
 * the interval vectors): ranks_bev / ranks_depth / ranks_feat sorted by voxel, ascending point
 * index inside a voxel (stable, where the reference's argsort is not); interval_starts /
 * interval_lengths; counts[0] = Np (kept points), counts[1] = Nv (intervals) — device ints, so that
 * the call itself never synchronises.  Bit-exact against the reference's vectors.
 *
 * ocrf_ht_prepare replaces get_sampling_point + fast_sample_prepare
 * (mmdet3d/models/necks/view_transformer_ocrf.py:687-740,785-852):
 *   ref_points (Z,n_pillars,3) device   get_reference_points_3d template (:651-673), normalised
 *   cams       (B*N,24) device          lidar2img 3x4 | img_aug 3x4 per camera-frame (get_projection, :675-685)
 *   pc_range_host  6 floats in HOST memory; w_in / h_in: network input size; depth0 / depth1:
 *   grid_config['depth'][0:2]; Wf, Hf, D: feature-map size and depth bins
 * outputs as above with ranks_bev = b*n_pillars + pillar; inside a pillar the order is (camera,
 * height) ascending = the reference's flattening order under a stable sort.
  * The workspace holds the look-back states of the single-launch prefix sums: one call at a time per workspace
 * (two calls that share it on different streams overwrite each other's states; a wave that waits ~a second for a
 * predecessor traps instead of spinning for ever).
 */
int ocrf_lss_prepare(int B, int N, int D, int H, int W, const float *frustum, const float *cams,
                     const float *grid_lower_host, const float *grid_interval_host, int gx, int gy, int gz,
                     int *ranks_bev, int *ranks_depth, int *ranks_feat, int *interval_starts,
                     int *interval_lengths, int *counts, void *workspace, size_t workspace_bytes,
                     ocrf_stream_t stream);
size_t ocrf_lss_prepare_workspace_bytes(int B, int N, int D, int H, int W, int gx, int gy, int gz);
int ocrf_ht_prepare(int B, int N, int Z, int n_pillars, int Wf, int Hf, int D, const float *ref_points,
                    const float *cams, const float *pc_range_host, float w_in, float h_in, float depth0,
                    float depth1, int *ranks_bev, int *ranks_depth, int *ranks_feat, int *interval_starts,
                    int *interval_lengths, int *counts, void *workspace, size_t workspace_bytes,
                    ocrf_stream_t stream);
size_t ocrf_ht_prepare_workspace_bytes(int B, int n_pillars);

/*
 * Per-forward calibration algebra on the device (the host glue of view_transformer.py:128-146,
 * view_transformer_ocrf.py:675-685 and :1135-1152 with data_utils.py:703-733): from the seven calibration
 * tensors of the reference's input tuple, where they already live, to the camera blocks the kernels above take —
 *   lss_block   (B*N,33)  inv(post_rots) | rots inv(intrins) | post_trans | trans | bda      (ocrf_lss_prepare)
 *   ht_block    (B*N,24)  lidar2img 3x4 | img_aug 3x4                        (ocrf_ht_prepare / ocrf_ht_project)
 *   camera_rows (B*N,36)  the `cameras` rows of ocrf_rasterize_forward for every camera-frame, built as the
 *                         reference builds its render camera (quirks included); needs c2w (B*N,4,4)
 * (any of the three may be NULL).  3x3 inverses and products are evaluated in double precision and rounded to
 * float once, so the values agree with the reference's float32 LAPACK / matmul chain to ~1 ulp, not bit for bit:
 * a pillar sample or frustum point that sits within that distance of a cell border can land in the neighbouring
 * cell (as it can between the reference's own CPU and CUDA runs).  No host read, no synchronisation.
 */
int ocrf_geometry_blocks(int B, int N, const float *rots, const float *trans, const float *intrins,
                         const float *post_rots, const float *post_trans, const float *bda, const float *c2w,
                         int H_in, int W_in, float znear, float zfar, float *lss_block, float *ht_block,
                         float *camera_rows, ocrf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Height-aware Opacity-based Attention (HOA) reductions
 * ------------------------------------------------------------------------------------------
 * All tensors float32, NCHW contiguous.
 *
 * ocrf_hoa_channel_stats + ocrf_hoa_opacity_mask_gate replace ObatinOpacityMask.forward
 * (mmdet3d/models/necks/view_transformer_ocrf.py:236-242) and `geom_feat * opacity_mask`
 * (:1197-1199):
 *   stats[b][0] = mean_c x[b][c], stats[b][1] = max_c x[b][c]                       (B,2,Y,X)
 *   mask  = sigmoid(conv2d(stats, conv_w (1,2,k,k), padding k/2, no bias) + opacity_bev)  (B,1,Y,X)
 *   gated = x * mask (skipped when gated == NULL)                                    (B,C,Y,X)
 */
int ocrf_hoa_channel_stats(const float *x, int B, int C, int Y, int X, float *stats,
                           ocrf_stream_t stream);
int ocrf_hoa_opacity_mask_gate(const float *x, const float *stats, const float *opacity_bev,
                               const float *conv_w, int k, int B, int C, int Y, int X, float *mask,
                               float *gated, ocrf_stream_t stream);

/*
 * Depthwise 3x3 convolution, padding 1, stride 1 (the first layer of OpacityVoxelToBEVConverter.conv_block,
 * view_transformer_ocrf.py:483-489) for the TRAINING path, where MIOpen's depthwise backward-weight costs
 * milliseconds on these 4..16-channel maps:
 *   ocrf_hoa_dw3x3:        y[b][c] = conv3x3(x[b][c], w[c] (9 floats)) + bias[c]   (bias may be NULL); also the
 *                          input gradient when called with dy and the flipped weights
 *   ocrf_hoa_dw3x3_wgrad:  partial[(b*C + c) * n_bands + band][10]: per 16-row band the nine sums
 *                          dy[y][x] * x[y+i-1][x+j-1] and the sum of dy; n_bands = ocrf_hoa_dw3x3_wgrad_bands(Y);
 *                          the caller adds the bands and samples (fixed order: deterministic).
 */
int ocrf_hoa_dw3x3(const float *x, const float *w, const float *bias, int B, int C, int Y, int X, float *y,
                   ocrf_stream_t stream);
int ocrf_hoa_dw3x3_wgrad_bands(int Y);
int ocrf_hoa_dw3x3_wgrad(const float *x, const float *dy, int B, int C, int Y, int X, float *partial,
                         ocrf_stream_t stream);

/*
 * HeightAttention.forward (view_transformer_ocrf.py:447-461) and its use `ca(x) * x`
 * (:499-514): the C channels are four height quarters of q = C/4 channels; per quarter g
 *   gate[b][g*q + o] = sigmoid( sum_h w2[g][o][h] * relu( sum_i w1[g][h][i] * max_{y,x} x[b][g*q+i] ) )
 * w1 is [4][hid][q] (convN.0.weight), w2 is [4][q][hid] (convN.2.weight), no biases.
 *   gate (B,C) is always written; gated = gate * x (B,C,Y,X) when gated != NULL.
 * C % 4 == 0, C <= 64, 4*hid <= 64.  workspace >= ocrf_hoa_height_attention_workspace_bytes(B, C).
 */
int ocrf_hoa_height_attention(const float *x, int B, int C, int hid, int Y, int X, const float *w1,
                              const float *w2, float *gate, float *gated, void *workspace,
                              size_t workspace_bytes, ocrf_stream_t stream);
size_t ocrf_hoa_height_attention_workspace_bytes(int B, int C);

/*
 * OpacityVoxelToBEVConverter (view_transformer_ocrf.py:463-518) as fused blocks, eval mode.
 * ocrf_hoa_unet_block = one `conv_block` (depthwise 3x3 + bias -> 1x1 + bias -> BatchNorm folded into
 * pw_w/pw_b by the caller -> ReLU, :485-491) over the virtual input
 *   mode 0: src0 (B,C0,H,W)                                   [encoder1, :498]
 *   mode 1: maxpool2x2(src0 (B,C0,2H,2W))                     [encoder2 / bottleneck, :501,:504]
 *   mode 2: cat(ConvTranspose2d_k2s2(src0 (B,C0,H/2,W/2); up_w (C0,Cup,2,2), up_b), src1 (B,C1,H,W))
 *                                                             [decoder2 / decoder1, :507-514]
 * with gate0 (B,C0) / gate1 (B,C1) (the producers' HeightAttention gates, NULL = none) multiplied in
 * while reading, `addend` (B,Cout,H,W) added after the ReLU (positional encoding, NULL = none), and
 * partial_max (B*Cout, ocrf_hoa_unet_tiles(H,W)) receiving per-tile channel maxima (NULL = skip).
 * Channel counts <= 16.  ocrf_hoa_height_gate_from_tiles turns those maxima into the HeightAttention
 * gate (B,C) (same w1/w2 layout as ocrf_hoa_height_attention); ocrf_hoa_gated_conv1x1 is the final
 * `output_conv` (:516) over gate * x: out (B,1,H,W) = bias[0] + sum_c w[c] * gate[b][c] * x[b][c].
 */
int ocrf_hoa_unet_block(const float *src0, const float *gate0, int C0, int H0, int W0, int mode,
                        const float *up_w, const float *up_b, int Cup, const float *src1,
                        const float *gate1, int C1, const float *dw_w, const float *dw_b,
                        const float *pw_w, const float *pw_b, int Cout, const float *addend, float *out,
                        float *partial_max, int B, int H, int W, ocrf_stream_t stream);
int ocrf_hoa_unet_tiles(int H, int W);
/* The whole converter of the architecture OcRFDet instantiates (input_channel = 13: 13 -> 4 -> 8 -> 16 -> 8 -> 4 -> 1,
 * ratio-1 HeightAttention gates, view_transformer_ocrf.py:463-518) as ONE call of six launches (csrc/hoa_v2b.hip:
 * latency-shaped forms of the five blocks above on 16 x 4 pixel tiles, each computing the gates of its producers in
 * its own prologue from the per-tile maxima they left, and the gated output conv; bit-identical to the block-wise
 * calls).  x (B,13,H,W), position (B,4,H,W), out (B,1,H,W); H and W multiples of 4.  `weights` (16-byte aligned):
 * ocrf_hoa_v2b_weights_len() floats — per block (encoder1, encoder2, bottleneck, decoder2, decoder1): dw_w (Cin,9),
 * dw_b (Cin), pw_w (Cout,Cin) and pw_b (Cout) with BatchNorm folded, the block's gate w1 (4,hid,q), w2 (4,q,hid);
 * then upconv2 w (16,8,2,2), b (8), upconv1 w (8,4,2,2), b (4), output conv w (4), b (1); zero-padded to a whole
 * number of 16-byte words (the kernels copy it to LDS with 16-byte loads).
 * workspace (16-byte aligned) >= ocrf_hoa_v2b_workspace_bytes(B, H, W) bytes (the five intermediates and their tile
 * maxima). */
int ocrf_hoa_v2b_weights_len(void);
size_t ocrf_hoa_v2b_workspace_bytes(int B, int H, int W);
int ocrf_hoa_v2b_forward(const float *x, const float *position, const float *weights, int B, int H, int W,
                         void *workspace, size_t workspace_bytes, float *out, ocrf_stream_t stream);
int ocrf_hoa_height_gate_from_tiles(int B, int C, int hid, int n_tiles, const float *partial_max,
                                    const float *w1, const float *w2, float *gate, ocrf_stream_t stream);
int ocrf_hoa_gated_conv1x1(const float *x, const float *gate, int B, int C, int H, int W, const float *w,
                           const float *bias, float *out, ocrf_stream_t stream);

/*
 * HOA-1 (view_transformer_ocrf.py:1159-1161): deformable cross attention between the Gaussian
 * opacity volume and the NeRF-branch alpha volume, eval mode, for the DeformableAttention2D that
 * OcRFDet instantiates (:639-648: dim 13, heads 1, dim_head 8, offset_groups 1, downsample_factor 4,
 * offset_kernel_size 6; mmdet3d/ops/cross_attention_2d.py:93-220):
 *   out (B,13,Y,X) = upsample(att(down(opacity), down(alpha)), (Y,X)) + opacity,
 * down/up = bilinear align_corners=True to/from (Y/6, X/6).  opacity, alpha: (B,13,Y,X).
 * weights: ocrf_hoa1_weights_len() floats packed as documented in csrc/hoa.hip (the Python module
 * packs its state_dict).  att_workspace: B*((13+8)*(Y/6)*(X/6) + 18*128) floats.  (Y/6)*(X/6) <= 1600.
 * Two kernel launches: the key / value tokens (one workgroup per group of kv tokens rebuilds the rows of q its
 * offset convolution reads), then attention + output projection + upsample + residual per 16x16 output tile.
 */
int ocrf_hoa1_forward(const float *opacity, const float *alpha, const float *weights, int B, int Y, int X,
                      float offset_scale, float *att_workspace, float *out, ocrf_stream_t stream);
int ocrf_hoa1_weights_len(void);

/*
 * Pillar projections consumed by the colour / alpha sampling and retain_valid_pixels
 * (view_transformer_ocrf.py:1057-1066 from get_sampling_point :687-740), same inputs as
 * ocrf_ht_prepare (pc_range: 6 floats in HOST memory):
 *   pix   (B,N,Z,n_pillars,2)  (u_norm*w_in, v_norm*h_in) of every sample, masked ones included
 *   mask  (B,N,Z,n_pillars)    bytes, the get_sampling_point mask
 *   voxel (B,Z,n_pillars,3)    metric voxel centres (the in-place scaled reference points, :690-692); may be NULL
 */
int ocrf_ht_project(int B, int N, int Z, int n_pillars, const float *ref_points, const float *cams,
                    const float *pc_range, float w_in, float h_in, float depth0, float depth1,
                    float *pix, unsigned char *mask, float *voxel, ocrf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Neck glue between the poolings, the render and HOA (eval mode; csrc/neck.hip)
 * ------------------------------------------------------------------------------------------ */

/*
 * OcRFViewTransformerFull.forward's pre-filter (view_transformer_ocrf.py:1323-1331) fused with the
 * channels-last permute the poolings need (:875, :901).  x (BN, D+2+C, HW) = DepthNet output:
 *   depth         (BN,D,HW)  softmax over the D depth logits (returned to the caller, :1334)
 *   filter_depth  (BN,D,HW)  depth, zeroed where depth < depth_threshold (= depth_threshold/D of the ctor)
 *   semantic      (BN,2,HW)  softmax over the 2 semantic logits
 *   feat_channels_last (BN,HW,C)  tran_feat * (semantic[:,1] >= semantic_threshold), i.e. the
 *                 (B,N,H,W,C) operand of bev_pool_v2
 * D <= 512.
 */
int ocrf_prefilter(const float *x, int BN, int D, int C, int HW, float depth_threshold,
                   float semantic_threshold, float *depth, float *filter_depth, float *semantic,
                   float *feat_channels_last, ocrf_stream_t stream);

/*
 * lidar_points_to_image_values + color_voxels' avg_color (view_transformer_ocrf.py:924-959):
 *   avg[b][q][c] = sum_{n: mask[b][n][q]} bilinear(imgs[b][n][c], pix[b][n][q]) / max(1, #valid n)
 * imgs (B,N,C,Hi,Wi) with C in {1,3}; pix (B,N,ZQ,2) PIXEL coordinates (x,y) — normalised as
 * (p/(size-1))*2-1 and sampled like F.grid_sample(bilinear, zeros, align_corners=True), :929-936;
 * mask (B,N,ZQ) bytes (torch.bool); avg (B,ZQ,C).  Hi/Wi are the extents the reference's view of the
 * image claims: for the alpha volume it passes the (H,W) stack viewed as (W,H) (:1123).
 */
int ocrf_pillar_sample_mean(const float *imgs, const float *pix, const unsigned char *mask, float *avg,
                            int B, int N, int C, int Hi, int Wi, int ZQ, ocrf_stream_t stream);

/*
 * retain_valid_pixels (view_transformer_ocrf.py:1004-1024): out = 255 except pixels
 * (clamp(trunc(pix.x)), clamp(trunc(pix.y))) of valid projections, which keep imgs' value.
 * imgs (B,N,C,H,W); cam_sel == NULL: every camera, out (B,N,C,H,W); cam_sel (B) int32 device
 * pointer: only camera cam_sel[b] of sample b (the one the caller consumes, :1104), out (B,C,H,W).
 */
int ocrf_retain_valid_pixels(const float *imgs, const float *pix, const unsigned char *mask,
                             const int *cam_sel, float *out, int B, int N, int C, int H, int W, int ZQ,
                             ocrf_stream_t stream);

/*
 * VoxelFeatureExtractor (view_transformer_ocrf.py:520-531, call :1051; Conv3d(1->Zh,k=1) +
 * BatchNorm3d folded by the caller into lift_a/lift_b + ReLU) fused with the four Gaussian heads
 * (:272-320, calls :1130-1133).  bev (B,C,YX) is the HT BEV feature, rgb_avg (B,Zh,YX,3) the
 * sampled voxel colour in 0..255 (divided by 255 inside, :1071).  Gaussian g = (b, h, q):
 *   f[c] = relu(lift_a[h]*bev[b][c][q] + lift_b[h]);  opacity (B,Zh*YX,1) = sigmoid(A_MLP(f)),
 *   scales (.,3) = softplus(S_MLP(f)), rotations (.,4) = normalize(R_MLP(f)), color (.,3) =
 *   sigmoid(C_MLP([f, rgb/255])).
 * params: ocrf_gauss_heads_params_len(C, Zh) floats packed as documented in csrc/neck.hip.
 * rotations must be 16-byte aligned, params 8-byte aligned; Zh in {1,2,4,6,8,13} (register tile).
 */
int ocrf_gauss_heads(const float *bev, const float *rgb_avg, const float *params, int B, int C, int Zh,
                     int YX, float *opacity, float *scales, float *rotations, float *color,
                     ocrf_stream_t stream);
int ocrf_gauss_heads_params_len(int C, int Zh);

/*
 * Backward of ocrf_gauss_heads: what autograd does for the reference through VoxelFeatureExtractor (:520-531) and the four
 * heads (:272-320, calls :1130-1133) on the (B,Zh,Y,X,C) voxel feature, here without that tensor (the lift and the hidden
 * units are recomputed from bev).  g_* are the gradients of the four outputs (same shapes; NULL = zero); d_bev (B,C,YX) is
 * fully written here; d_params receives the gradient w.r.t. every entry of `params`, in the layout of `params`
 * (lift_a / lift_b first: the caller owns the folding of Conv3d + BatchNorm3d — batch statistics in training — and
 * differentiates through it).  No gradient for rgb_avg (sampled from the camera images, :1071).  Deterministic.
 * workspace: ocrf_gauss_heads_backward_workspace_bytes(B, C, Zh, YX) bytes (0: unsupported shape).
 */
size_t ocrf_gauss_heads_backward_workspace_bytes(int B, int C, int Zh, int YX);
int ocrf_gauss_heads_backward(const float *bev, const float *rgb_avg, const float *params, int B, int C, int Zh, int YX,
                              const float *g_opacity, const float *g_scales, const float *g_rotations,
                              const float *g_color, float *d_bev, float *d_params, void *workspace,
                              size_t workspace_bytes, ocrf_stream_t stream);

/*
 * NeRF branch (view_transformer_ocrf.py:1094-1121) from z (M,32,h2,w2) = ResizeNetwork.conv2's
 * output (:546); the caller composes upsample2 (k2 s2) -> upsample3 (k4 s4) -> first Linear of each
 * consumer into per-sub-position maps (64 positions of the 8x8 up-sampling cell):
 *   ocrf_nerf_alpha:  alpha (M,8*h2,8*w2) = 1 - exp(-softplus(w_sigma[:,pos] . z + c_sigma[pos]))
 *                     (sigma = Softplus(Linear(Linear(.))), :605, :1099-1102)
 *   ocrf_nerf_render: for sample b and camera cam_sel[b] (int32 device vector), with sparse_rgb
 *                     (B,3,H,W) from ocrf_retain_valid_pixels: render_image_n (B,3,H,W) =
 *                     alpha * relu(img_feat_resize1) * softmax(C_MLP_nerf), render_depth_n (B,1,H,W) =
 *                     alpha * relu(img_feat_resize2)   (T == 1 and the depth weight == 1, :1108-1117).
 * params of ocrf_nerf_render: ocrf_nerf_render_params_len() floats, layout in csrc/neck.hip.
 */
int ocrf_nerf_alpha(const float *z, const float *w_sigma, const float *c_sigma, float *alpha, int M,
                    int h2, int w2, ocrf_stream_t stream);
int ocrf_nerf_render(const float *z, const int *cam_sel, const float *alpha, const float *sparse_rgb,
                     const float *params, int B, int N, int h2, int w2, float *render_image_n,
                     float *render_depth_n, ocrf_stream_t stream);
int ocrf_nerf_render_params_len(void);

/*
 * DualFeatFusion.forward (view_transformer_ocrf.py:203-213; MS_CAM :36-66), eval mode, one pass:
 *   out = cf * x1 + (1 - cf) * x2,  cf = sigmoid(local_att([x1; x2]) + global_vec[b]),
 *   local_att = Conv1x1(2C->M) + BN + ReLU + Conv1x1(M->C) + BN with both BatchNorms folded by the
 *   caller into params = W1t[2C][M] | b1[M] | W2[C][M] | b2[C]; global_vec (B,C) is MS_CAM's
 *   global_att branch of the pooled map (a (B,2C) vector through two tiny layers: the caller's).
 * x1, x2, out: (B,C,YX).  (C, M) in {(80, 40), (64, 32)}; params 8-byte aligned.
 */
int ocrf_dual_feat_fusion(const float *x1, const float *x2, const float *params, const float *global_vec,
                          float *out, int B, int C, int M, int YX, ocrf_stream_t stream);
/* ... and, in the same pass, out_plus = addend + out (addend, out_plus (B,C,YX)): the fused map with the positional
 * encoding ProbNet's first convolution reads it with (view_transformer_ocrf.py:1188) — one more store per element
 * instead of an elementwise launch over the map. */
int ocrf_dual_feat_fusion_plus(const float *x1, const float *x2, const float *params, const float *global_vec,
                               float *out, const float *addend, float *out_plus, int B, int C, int M, int YX,
                               ocrf_stream_t stream);

/*
 * CBAM / ProbNet tail (view_transformer_ocrf.py:68-137 ChannelAttention, SpatialAttention, ResCBAMBlock;
 * :139-201 ProbNet; MS_CAM.global_att :50-58), eval mode.  The 3x3 convolutions stay with MIOpen; these four
 * entry points replace the ~30 elementwise / reduction launches between them.
 *
 * ocrf_plane_bias_act_stats: one pass over y (B,C,YX), in place when write != 0: y += bias[c] (bias may be
 *   NULL; BatchNorm folded by the caller), ReLU when relu != 0; when psum != NULL also S partial sums and
 *   maxima of the result per plane -> psum / pmax [(b*out_C + c_off + c)*S + s] (two inputs can fill one
 *   (B, out_C, S) block: MS_CAM pools cat(x1, x2)).  y 16-byte aligned.
 * ocrf_channel_mlp: v_mean = sum_s psum * inv_n, v_max = max_s pmax;
 *   out[b][n] = act(W2.relu(W1.v_mean + b1) + b2 [+ W2.relu(W1.v_max + b1) + b2]), act = sigmoid or identity;
 *   W1 (M,K), W2 (N,M) row-major, b1 / b2 may be NULL; K <= 256, M <= 64.
 * ocrf_scaled_channel_stats: stats (B,2,YX) = mean_c, max_c of scale[b][c] * x[b][c] (SpatialAttention's input
 *   of the channel-gated map, which is never materialised).
 * ocrf_cbam_tail: m = sigmoid(conv_kxk(stats)); o_c = relu(m * (scale_c * y_c) + res_c);
 *   logit (B,YX) = sum_c wm_c * o_c + bm (ProbNet.mask_net); block_out (B,C,YX) receives o when not NULL.
 */
int ocrf_plane_bias_act_stats(float *y, const float *bias, int B, int C, int YX, int relu, int write, int S,
                              int out_C, int c_off, float *psum, float *pmax, ocrf_stream_t stream);
/* the statistics of TWO tensors y1 (B,C1,YX), y2 (B,C2,YX) as if concatenated along the channels, in ONE launch:
 * psum / pmax (B, C1 + C2, S) (MS_CAM's global branch pools cat(x1, x2) :50-58; 16-byte aligned inputs) */
int ocrf_plane_stats_pair(const float *y1, const float *y2, int B, int C1, int C2, int YX, int S, float *psum,
                          float *pmax, ocrf_stream_t stream);
int ocrf_channel_mlp(const float *psum, const float *pmax, int B, int K, int S, float inv_n, const float *W1,
                     const float *b1, const float *W2, const float *b2, int M, int N, int use_max, int do_sigmoid,
                     float *out, ocrf_stream_t stream);
int ocrf_scaled_channel_stats(const float *x, const float *scale, int B, int C, int YX, float *stats,
                              ocrf_stream_t stream);
int ocrf_cbam_tail(const float *y, const float *scale, const float *stats, const float *conv_w, int k,
                   const float *res, const float *wm, float bm, int B, int C, int Y, int X, float *logit,
                   float *block_out, ocrf_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Per-kernel device timer (measurement aid for bench.py; not part of the reference's surface)
 * ------------------------------------------------------------------------------------------
 * While a timer is armed for kernel id K, every launch of K inside the library is bracketed by
 * a hipEvent pair recorded on the launch stream (hipExtLaunchKernelGGL), up to `capacity`
 * launches.  ocrf_timer_read waits for the recorded pairs and returns their elapsed
 * milliseconds.  Several timers (for different kernel ids) may be armed at once; arming with
 * timer == NULL disarms all of them.  Host-side only. */
enum {
  OCRF_K_BEV_POOL_FWD = 1,      /* bev_pool_tile_kernel */
  OCRF_K_BEV_POOL_FIXUP = 2,    /* (retired) */
  OCRF_K_BEV_POOL_INTERVAL = 3, /* bev_pool_interval_kernel */
  OCRF_K_BEV_POOL_GRAD = 4,     /* bev_pool_grad_vec_kernel */
  OCRF_K_BEV_POOL_NCHW = 5,     /* (retired: the tile kernel writes the final layout itself) */
  OCRF_K_BEV_POOL_MFMA = 6,     /* bev_pool_mfma_kernel<C / 16> */
  OCRF_K_BEV_POOL_PANEL = 7,    /* bev_pool_panel_kernel<C / 4> */
  OCRF_K_BEV_POOL_CELL_WEIGHTS = 8, /* bev_pool_cell_weights_kernel */
  OCRF_K_RASTER_PREPROCESS = 10, /* raster_preprocess_kernel */
  OCRF_K_RASTER_BLEND = 11,      /* raster_blend_kernel */
  OCRF_K_RASTER_GATHER = 12,     /* raster_scatter_kernel */
  OCRF_K_RASTER_SCAN = 13,       /* raster_bucket_scan_kernel */
  OCRF_K_RASTER_BLEND_BWD = 15,  /* raster_blend_kernel<false, true> */
  OCRF_K_RASTER_PRE_BWD = 16,    /* raster_preprocess_backward_kernel */
  OCRF_K_RASTER_PLAN_UPDATE = 17,  /* raster_plan_head_kernel (rounds 3-4: raster_plan_update_kernel) */
  OCRF_K_RASTER_BLEND_SORTED = 18, /* raster_blend_sorted_kernel<*>, first pass */
  OCRF_K_RASTER_BLEND_SECOND = 19, /* raster_blend_sorted_kernel<*>, second pass (tile pairs handed over by the first) */
  OCRF_K_HOA_STATS = 20,         /* hoa_channel_stats_kernel */
  OCRF_K_HOA_MASK_GATE = 21,     /* hoa_mask_gate_kernel */
  OCRF_K_HOA_HEIGHT_MAX = 22,    /* hoa_height_max_kernel */
  OCRF_K_HOA_HEIGHT_GATE = 23,   /* hoa_height_gate_kernel / hoa_height_gate_from_tiles_kernel */
  OCRF_K_HOA_UNET_BLOCK = 24,    /* hoa_unet_block_kernel */
  OCRF_K_HOA_OUT_CONV = 25,      /* hoa_gated_conv1x1_kernel */
  OCRF_K_HOA1_ATTN = 26,         /* hoa1_attention_upsample_kernel */
  OCRF_K_HOA1_UP = 27,           /* unused since round 2 (fused into OCRF_K_HOA1_ATTN) */
  OCRF_K_HOA1_Q = 28,            /* unused since round 2 (q is rebuilt by its consumers) */
  OCRF_K_HOA1_KV = 29,           /* hoa1_kv_kernel */
  OCRF_K_HOA_DW3X3 = 30,         /* hoa_dw3x3_kernel */
  OCRF_K_HOA_DW3X3_WGRAD = 31,   /* hoa_dw3x3_wgrad_kernel */
  OCRF_K_LSS_KEYS = 40,          /* lss_keys_hist_kernel (keys + the first radix histogram) */
  OCRF_K_RADIX_HIST = 41,        /* radix_hist_kernel */
  OCRF_K_SCAN = 42,              /* scan_apply_kernel<T> */
  OCRF_K_RADIX_SCATTER = 43,     /* radix_scatter_kernel<*> */
  OCRF_K_LSS_BOUNDS = 44,        /* lss_intervals_kernel */
  OCRF_K_LSS_EMIT = 45,          /* unused since round 5 (the last radix pass scatters into the rank vectors) */
  OCRF_K_HT_COUNT = 46,          /* ht_valid_kernel */
  OCRF_K_HT_EMIT = 47,           /* ht_emit_kernel */
  OCRF_K_HT_PROJECT = 48,        /* ht_project_kernel */
  OCRF_K_NECK_PREFILTER = 60,    /* neck_prefilter_kernel */
  OCRF_K_NECK_SAMPLE = 61,       /* neck_pillar_sample_mean_kernel<C> */
  OCRF_K_NECK_RETAIN = 62,       /* neck_fill_kernel + neck_retain_scatter_kernel */
  OCRF_K_NECK_HEADS = 63,        /* neck_gauss_heads_kernel */
  OCRF_K_NECK_NERF_ALPHA = 64,   /* neck_nerf_alpha_kernel */
  OCRF_K_NECK_NERF_RENDER = 65,  /* neck_nerf_render_kernel */
  OCRF_K_NECK_FUSION = 66,       /* neck_dual_fusion_kernel<C, M> */
  OCRF_K_NECK_PLANE_PASS = 67,   /* neck_plane_pass_kernel<WRITE, RELU, STATS> */
  OCRF_K_NECK_CHANNEL_MLP = 68,  /* neck_channel_mlp_kernel */
  OCRF_K_NECK_SCALED_STATS = 69, /* neck_scaled_channel_stats_kernel */
  OCRF_K_NECK_CBAM_TAIL = 70,    /* neck_cbam_tail_kernel */
  OCRF_K_RASTER_SH = 50,         /* sh_colors_kernel */
  OCRF_K_RASTER_SH_BWD = 51,     /* sh_colors_backward_kernel */
  OCRF_K_NECK_HEADS_BWD = 71,    /* neck_gauss_heads_backward_kernel */
  OCRF_K_NECK_HEADS_BWD_SUM = 72 /* neck_partial_rows_sum_kernel (both stages) */
};
const char *ocrf_kernel_name(int kernel_id);           /* symbol as rocprofv3 prints it */
/* Diagnostic: a one-thread kernel that stores the device's constant-rate clock (wall_clock64, 100 MHz)
 * into *slot (device) when the stream reaches it — timelines inside a hipGraph replay, where host
 * events cannot be placed and the profiler's per-kernel signals perturb the overlap. */
int ocrf_diag_stamp(unsigned long long *slot, void *stream);
/* Diagnostic: n_blocks one-wave workgroups that each record where they ran: out[b] = XCC_ID << 16 | HW_ID[15:0]
 * (CU_ID bits 11:8, SH_ID 12, SE_ID 15:13) after idling for spin_ticks of the 100 MHz clock — the compute units a
 * CU-masked stream really owns. */
/* Diagnostic: with a device buffer of (tile pairs x items x 4 waves x 8) u64 set, the next ocrf_rasterize_planned
 * calls run an instrumented build of the sorted blend: per wave scan / stage / blend cycles (s_memtime) and the
 * number of scanned, staged, listed and evaluated records.  NULL switches it off. */
int ocrf_diag_plan_stats(unsigned long long *buf);
int ocrf_diag_plan_resident(void); /* workgroups the persistent sorted blend launches (occupancy API x CUs) */
int ocrf_diag_where(int n_blocks, unsigned *out, int spin_ticks, void *stream);
/* HIP streams for the hot path's chains (host pointers): cu_mask (n_words x 32 bits, NULL = every CU) restricts the
 * stream's kernels to those compute units (hipExtStreamCreateWithCUMask); without a mask `priority` is the HIP stream
 * priority (lower = more urgent).  The stream is non-blocking w.r.t. the legacy default stream. */
/* Node census of a captured hipGraph (hipGraph_t; host pointers).  Graphs with memset nodes fault on replay after an
 * intervening hipMemcpyAsync on ROCm 7.2 / gfx950: this library zero-fills with a kernel, and owners of captured graphs
 * (GraphedNeck) refuse a graph whose census shows a memset node (e.g. a torch.zeros added inside the captured region). */
int ocrf_graph_node_census(void *graph, int *n_kernel, int *n_memset, int *n_memcpy, int *n_other);
int ocrf_stream_create(const uint32_t *cu_mask, int n_words, int priority, void **stream_out);
/*
 * One host call per step (csrc/step.hip).  The reference's boundary is one pybind call per op
 * (mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:30-57); a step of the hot path here is ~18 launches behind seven of the
 * entry points above on two HIP streams, and issuing them from Python costs the host as long as the device needs to
 * run them.  An ocrf_step holds the calls of ONE step — entry point + argument values, recorded once while the step ran
 * eagerly — and the fork / join points between its streams; ocrf_hotpath_step replays them from C on the caller's
 * streams: the same entry points with the same arguments, launched eagerly (no hipGraph).  Pointer arguments are used as
 * recorded: the caller keeps inputs, outputs and scratch alive and in place, as for a captured graph.
 *   ocrf_step_fn_id(name)    id of an entry point a step can hold (-1: not supported — issue the step call by call):
 *                            ocrf_bev_pool_v2_nchw_planned / _mfma / _panel, ocrf_bev_pool_cell_weights,
 *                            ocrf_rasterize_planned, ocrf_raster_plan_build, ocrf_hoa1_forward, ocrf_hoa_v2b_forward,
 *                            ocrf_hoa_channel_stats, ocrf_hoa_opacity_mask_gate, ocrf_stream_write_value32
 *   ocrf_step_add_call       args: the entry point's arguments WITHOUT its trailing stream, one 64-bit word each —
 *                            pointers / sizes / ints (sign-extended) as such, a float as its bit pattern in the low word;
 *                            slot: which of the streams handed to the replay the call goes to (< 8)
 *   ocrf_step_add_fork/_join stream slot `to` continues only after what slot `from` holds so far (an event record +
 *                            wait; the events belong to the step object)
 *   ocrf_step_run            replay on `streams[0 .. n_streams)`; ocrf_hotpath_step: slot 0 = main, slot 1 = side.
 * All host pointers; not thread-safe per object; returns the first non-zero status of a replayed call.
 */
typedef struct ocrf_step ocrf_step;
int ocrf_step_fn_id(const char *name);
int ocrf_step_fn_args(int fn);
int ocrf_step_create(ocrf_step **out);
void ocrf_step_destroy(ocrf_step *step);
int ocrf_step_add_call(ocrf_step *step, int fn, int slot, int n_args, const uint64_t *args);
int ocrf_step_add_fork(ocrf_step *step, int from, int to);
int ocrf_step_add_join(ocrf_step *step, int from, int to);
int ocrf_step_size(const ocrf_step *step);
int ocrf_step_run(ocrf_step *step, const ocrf_stream_t *streams, int n_streams);
int ocrf_hotpath_step(ocrf_step *step, ocrf_stream_t main_stream, ocrf_stream_t side_stream);

/* value -> *ptr (device int) in stream order, no kernel launch (hipStreamWriteValue32). */
int ocrf_stream_write_value32(int *ptr, int value, ocrf_stream_t stream);
int ocrf_stream_destroy(void *stream);
int ocrf_timer_create(int capacity, void **timer_out); /* host pointers */
int ocrf_timer_arm(void *timer, int kernel_id);
int ocrf_timer_read(void *timer, float *ms_out /* host */, int capacity, int *count_out /* host */);
int ocrf_timer_destroy(void *timer);

/* Diagnostics of the pooling kernel (A/B timing of variants in one process; never on the product path).
 * ocrf_tune_set: key 0 = rounds of the point walk per slice of a heavy tile (1..64; default 4), key 1 = deal
 * workgroups to XCDs in contiguous unit ranges (default 1), key 2 = cap on the workgroups of the launch (0 = one per
 * unit), key 3 = voxels per tile (32 | 64; default 64).  Plans and workspaces are sized for the values in
 * force when they were built.  ocrf_diag_bev_pool_stamps: the planned pooling with s_memtime accumulated per
 * phase and unit into stamps[ocrf_bev_pool_max_units(...)][8] = {table + zero-fill, staging, gather, combine,
 * write-out, points of the unit, -, -}. */
int ocrf_tune_set(int key, int value);
int ocrf_bev_pool_max_units(int c, int n_points, int B, int Z, int Y, int X);
int ocrf_diag_bev_pool_stamps(int c, int n_points, const float *depth, const float *feat, const int *ranks_depth,
                              const int *ranks_feat, void *plan, float *out, int B, int Z, int Y, int X, int layout,
                              void *workspace, unsigned long long *stamps, ocrf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OCRF_HIP_H */
