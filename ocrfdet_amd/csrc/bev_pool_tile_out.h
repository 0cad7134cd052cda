// How a finished 8 x 8-voxel tile of the panel poolings (bev_pool_mfma.hip, bev_pool_panel.hip) leaves the chip.
//
// The tile's 64 x C pooled rows sit in LDS (`tile`, pitch ldq float4).  A tile whose panels were split over several
// units reduces first: every unit writes its partial tile as a write-through slab (sc1 stores: in memory once the
// wave's vmcnt drains — cdna_hip_programming.md 'In-launch split-K reduction'), takes a ticket, and the LAST arriver
// adds the slabs in slice order (fixed order whoever arrives last: no float atomics, bitwise reproducible) and writes
// the tile in the caller's layout.  Returns after the tile (or the slab) is on its way; every thread of the workgroup
// calls it.
#pragma once
#include <hip/hip_runtime.h>

namespace pool_out {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kTV = 64;          // voxels per tile
constexpr int kTS = 8;           // tile side: voxel slot v = 8 (y - y0) + (x - x0)
constexpr int kWaves = 4;        // 256 threads

struct Dest {
  int C, Y, X, Z, layout;        // layout 0: (B,C,Z,Y,X)  1: (B,Z*C,Y,X)  2: rows (n_vox, C)
  float* out;
  float4* slabs;
  int* arrive;                   // [n_tiles], zero between calls
};

// tile: LDS [64][ldq]; s_flag: one LDS int; slab_first: first slab of the tile (n_slices > 1)
template <int C4>
__device__ __forceinline__ void leave(float4* tile, int* s_flag, const Dest& d, int tileid, int plane, int y0, int x0,
                                      int n_slices, int slice, int slab_first) {
  constexpr int ldq = C4 | 1;
  constexpr int C = 4 * C4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int gpw = 64 / C4, gpb = gpw * kWaves;
  const int gi = lane / C4, lg = lane % C4, gb = wave * gpw + gi;
  const long YX = (long)d.Y * d.X;
  if (n_slices > 1) {
    float4* slab = d.slabs + (long)(slab_first + slice) * kTV * C4;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, kTV * C4 * (int)sizeof(float4), 0x00020000);
    if (gi < gpw) {
      for (int v = gb; v < kTV; v += gpb) {
        const float4 x = tile[v * ldq + lg];
        const u32x4 bits = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (v * C4 + lg) * (int)sizeof(float4), 0, 16);      // aux 16 = sc1
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(&d.arrive[tileid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_flag = (old == n_slices - 1) ? 1 : 0;
    }
    __syncthreads();
    if (*s_flag == 0) return;
    // the ticket counter is back at 0 for the next call (plans keep it across calls)
    if (tid == 0) __hip_atomic_store(&d.arrive[tileid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // system-scope loads (sc0 sc1: served past this XCD's L2, where the other units' write-through stores are)
    // instead of an acquire fence, whose buffer_inv would throw away the L2 lines the XCD's other workgroups gather from
    float4* s0 = d.slabs + (long)slab_first * kTV * C4;
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(s0, 0, n_slices * kTV * C4 * (int)sizeof(float4), 0x00020000);
    if (gi < gpw) {
      // slice order for the sums, eight independent slab loads in flight per lane
      constexpr int kSB = 8;
      for (int v = gb; v < kTV; v += gpb) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sb = 0; sb < n_slices; sb += kSB) {
          u32x4 r[kSB];
#pragma unroll
          for (int k = 0; k < kSB; ++k) {
            const int s = min(sb + k, n_slices - 1);
            r[k] = __builtin_amdgcn_raw_buffer_load_b128(srs, ((s * kTV + v) * C4 + lg) * (int)sizeof(float4), 0, 17);
          }
#pragma unroll
          for (int k = 0; k < kSB; ++k) {
            if (sb + k < n_slices) {
              const float4 x = make_float4(__uint_as_float(r[k].x), __uint_as_float(r[k].y), __uint_as_float(r[k].z),
                                           __uint_as_float(r[k].w));
              sum = (sb + k == 0) ? x : make_float4(sum.x + x.x, sum.y + x.y, sum.z + x.z, sum.w + x.w);
            }
          }
        }
        tile[v * ldq + lg] = sum;
      }
    }
    __syncthreads();
  }
  if (d.layout == 2) {
    float4* o = reinterpret_cast<float4*>(d.out);
    if (gi < gpw)
      for (int v = gb; v < kTV; v += gpb) {
        const int y = y0 + (v >> 3), x = x0 + (v & 7);
        if (y < d.Y && x < d.X) o[((long)plane * YX + (long)y * d.X + x) * C4 + lg] = tile[v * ldq + lg];
      }
  } else {
    const int b = plane / d.Z, z = plane % d.Z;
    const long base0 = (d.layout == 0) ? (((long)b * C * d.Z + z) * YX) : ((long)plane * C * YX);
    const long cstride = (d.layout == 0) ? (long)d.Z * YX : YX;
    // a wave instruction writes eight 32-byte runs (the tile's rows) of one channel plane
    const int y = y0 + (lane >> 3), x = x0 + (lane & 7);
    if (y < d.Y && x < d.X) {
      float* o = d.out + base0 + (long)y * d.X + x;
      for (int q = wave; q < C4; q += kWaves) {
        const float4 t = tile[lane * ldq + q];
        float* oc = o + (long)(4 * q) * cstride;
        oc[0] = t.x;
        oc[cstride] = t.y;
        oc[2 * cstride] = t.z;
        oc[3 * cstride] = t.w;
      }
    }
  }
}

}  // namespace pool_out
