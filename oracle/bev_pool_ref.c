/*
 * ORACLE — test infrastructure only.  Nothing under ocrfdet_amd/ may import, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * CPU restatement of the reference's BEVPoolv2 voxel pooling:
 *   forward : mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48  (bev_pool_v2_kernel)
 *   backward: mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:67-121 (bev_pool_grad_kernel)
 * One loop iteration here == one CUDA thread there; the arithmetic inside an iteration is in
 * the same order (sequential fp32 accumulation in list order).  nvcc contracts `a += b*c` into
 * an FMA by default, so the accumulation is written with fmaf().
 *
 * Pinned by: the reference's known-answer test (bev_pool.py:145-176) and the identity
 * bev_pool_v2 == index_add(depth[rd]*feat[rf], rb) — see tests/test_oracle_bev_pool.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* bev_pool_cuda.cu:21-48.  `out` must be pre-zeroed by the caller (bev_pool.py:27). */
void oracle_bev_pool_v2(int c, int n_intervals, const float *depth, const float *feat,
                        const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                        const int *interval_starts, const int *interval_lengths, float *out)
{
    const long total = (long)n_intervals * c;
#pragma omp parallel for schedule(dynamic, 1024)
    for (long idx = 0; idx < total; idx++) {
        int index = (int)(idx / c);          /* :31 */
        int cur_c = (int)(idx % c);          /* :32 */
        int interval_start = interval_starts[index];
        int interval_length = interval_lengths[index];
        float psum = 0.f;
        for (int i = 0; i < interval_length; i++) {       /* :39-43 */
            const float d = depth[ranks_depth[interval_start + i]];
            const float f = feat[(long)ranks_feat[interval_start + i] * c + cur_c];
            psum = fmaf(f, d, psum);
        }
        out[(long)ranks_bev[interval_start] * c + cur_c] = psum;   /* :45-47 */
    }
}

/* bev_pool_cuda.cu:67-121.  Intervals here are runs of equal ranks_feat (bev_pool.py:47-57). */
void oracle_bev_pool_v2_grad(int c, int n_intervals, const float *out_grad, const float *depth,
                             const float *feat, const int *ranks_depth, const int *ranks_feat,
                             const int *ranks_bev, const int *interval_starts,
                             const int *interval_lengths, float *depth_grad, float *feat_grad)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int idx = 0; idx < n_intervals; idx++) {
        int interval_start = interval_starts[idx];
        int interval_length = interval_lengths[idx];
        for (int i = 0; i < interval_length; i++) {       /* :91-105 */
            const float *og = out_grad + (long)ranks_bev[interval_start + i] * c;
            const float *ft = feat + (long)ranks_feat[interval_start + i] * c;
            float grad_sum = 0.f;
            for (int cur_c = 0; cur_c < c; cur_c++)
                grad_sum = fmaf(og[cur_c], ft[cur_c], grad_sum);
            depth_grad[ranks_depth[interval_start + i]] = grad_sum;   /* plain store :103-104 */
        }
        for (int cur_c = 0; cur_c < c; cur_c++) {          /* :109-120 */
            float grad_sum = 0.f;
            for (int i = 0; i < interval_length; i++) {
                const float og = out_grad[(long)ranks_bev[interval_start + i] * c + cur_c];
                const float d = depth[ranks_depth[interval_start + i]];
                grad_sum = fmaf(og, d, grad_sum);
            }
            feat_grad[(long)ranks_feat[interval_start] * c + cur_c] = grad_sum;
        }
    }
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* test / bench infrastructure: the OpenMP team size of every oracle entry point (1 = the scalar port) */
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
