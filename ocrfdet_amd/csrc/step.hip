// One host call per step: a recorded list of this library's own C-ABI calls, replayed from C.
//
// The reference launches one pybind call per op (mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:30-57); a step of the hot
// path here is ~18 kernel launches behind 7 entry points on two HIP streams, and issued from Python (ctypes argument
// conversion, torch stream / event objects) it costs the host 225 us — as long as the device needs to run it.  A step
// object holds the calls of one step (entry point + argument values, recorded once by ocrfdet_amd._lib.StepRecorder
// while the step ran eagerly) and the fork / join points of its streams; ocrf_hotpath_step replays them: the same
// entry points, the same arguments, the caller's streams.  Nothing is captured into a hipGraph (replaying the step as a
// graph was measured slower: 0.28 vs 0.23 ms) — the launches are issued eagerly, only without the interpreter.
//
// Every pointer argument is used as recorded: the caller keeps the tensors alive and in place (persistent inputs,
// outputs and scratch), exactly as for a captured graph.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "ocrf_hip.h"

namespace {

enum Fn : int {
  kPoolPlanned = 0,
  kPoolMfma,
  kPoolCellWeights,
  kPoolPanel,
  kRasterizePlanned,
  kHoa1,
  kHoaV2b,
  kHoaStats,
  kHoaMaskGate,
  kStreamWrite32,
  kRasterPlanBuild,
  kRasterizeForward,
  kPoolDyn,
  kLssPrepare,
  kHtPrepare,
  kGeometryBlocks,
  kRasterizeForwardSets,
  kFnCount
};

struct FnInfo {
  const char* name;
  int n_args;          // without the trailing stream
};

const FnInfo kFns[kFnCount] = {
    {"ocrf_bev_pool_v2_nchw_planned", 17}, {"ocrf_bev_pool_v2_nchw_mfma", 19},  {"ocrf_bev_pool_cell_weights", 9},
    {"ocrf_bev_pool_v2_nchw_panel", 20},   {"ocrf_rasterize_planned", 39},      {"ocrf_hoa1_forward", 9},
    {"ocrf_hoa_v2b_forward", 9},           {"ocrf_hoa_channel_stats", 6},       {"ocrf_hoa_opacity_mask_gate", 11},
    {"ocrf_stream_write_value32", 2},      {"ocrf_raster_plan_build", 12},      {"ocrf_rasterize_forward", 23},
    {"ocrf_bev_pool_v2_nchw_dyn", 19},     {"ocrf_lss_prepare", 20},            {"ocrf_ht_prepare", 22},
    {"ocrf_geometry_blocks", 16},          {"ocrf_rasterize_forward_sets", 24},
};

constexpr int kMaxArgs = 40;
constexpr int kMaxStreams = 8;

struct Cmd {
  int kind;            // 0: call, 1: fork (from -> to), 2: join (`to` waits for `from`)
  int fn, slot, from, to;
  uint64_t a[kMaxArgs];      // ints / pointers / sizes as 64-bit words, floats as their bit pattern
};

// typed views of a recorded argument word
template <typename T>
inline T P(uint64_t v) { return reinterpret_cast<T>(static_cast<uintptr_t>(v)); }
inline int I(uint64_t v) { return static_cast<int>(static_cast<int64_t>(v)); }
inline long L(uint64_t v) { return static_cast<long>(static_cast<int64_t>(v)); }
inline size_t Z(uint64_t v) { return static_cast<size_t>(v); }
inline float F(uint64_t v) {
  const uint32_t b = static_cast<uint32_t>(v);
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

int call(const Cmd& c, ocrf_stream_t s) {
  const uint64_t* a = c.a;
  switch (c.fn) {
    case kPoolPlanned:
      return ocrf_bev_pool_v2_nchw_planned(I(a[0]), I(a[1]), P<const float*>(a[2]), P<const float*>(a[3]),
                                           P<const int*>(a[4]), P<const int*>(a[5]), P<void*>(a[6]), P<float*>(a[7]),
                                           I(a[8]), I(a[9]), I(a[10]), I(a[11]), I(a[12]), P<void*>(a[13]), Z(a[14]), Z(a[15]),
                                           Z(a[16]), s);
    case kPoolMfma:
      return ocrf_bev_pool_v2_nchw_mfma(I(a[0]), I(a[1]), P<const int*>(a[2]), P<const int*>(a[3]), P<const int*>(a[4]),
                                        P<const int*>(a[5]), P<const int*>(a[6]), P<const int*>(a[7]),
                                        P<const int*>(a[8]), P<const float*>(a[9]), P<const float*>(a[10]),
                                        P<float*>(a[11]), I(a[12]), I(a[13]), I(a[14]), I(a[15]), I(a[16]),
                                        P<int*>(a[17]), P<void*>(a[18]), s);
    case kPoolCellWeights:
      return ocrf_bev_pool_cell_weights(I(a[0]), P<const int*>(a[1]), P<const int*>(a[2]), P<float*>(a[3]), I(a[4]),
                                        P<const int*>(a[5]), P<const int*>(a[6]), P<float*>(a[7]),
                                        P<const float*>(a[8]), s);
    case kPoolPanel:
      return ocrf_bev_pool_v2_nchw_panel(I(a[0]), I(a[1]), P<const int*>(a[2]), P<const int*>(a[3]), P<const int*>(a[4]),
                                         P<const int*>(a[5]), P<const int*>(a[6]), P<const int*>(a[7]),
                                         P<const unsigned short*>(a[8]), P<const float*>(a[9]), P<const float*>(a[10]),
                                         P<float*>(a[11]), I(a[12]), I(a[13]), I(a[14]), I(a[15]), I(a[16]),
                                         P<int*>(a[17]), P<void*>(a[18]), Z(a[19]), s);
    case kRasterizePlanned:
      return ocrf_rasterize_planned(P<const void*>(a[0]), Z(a[1]), I(a[2]), I(a[3]), L(a[4]), I(a[5]), I(a[6]), I(a[7]),
                                    I(a[8]), P<const int*>(a[9]), P<const float*>(a[10]), P<const float*>(a[11]),
                                    P<const float*>(a[12]), F(a[13]), P<const float*>(a[14]), P<const float*>(a[15]),
                                    I(a[16]), P<float*>(a[17]), P<float*>(a[18]), P<float*>(a[19]), P<int*>(a[20]),
                                    P<int*>(a[21]), P<void*>(a[22]), Z(a[23]), I(a[24]), P<const float*>(a[25]),
                                    P<void*>(a[26]), Z(a[27]), I(a[28]), P<const int*>(a[29]), I(a[30]),
                                    P<const float*>(a[31]), I(a[32]), P<const void*>(a[33]), Z(a[34]), I(a[35]), I(a[36]),
                                    L(a[37]), P<int*>(a[38]), s);
    case kHoa1:
      return ocrf_hoa1_forward(P<const float*>(a[0]), P<const float*>(a[1]), P<const float*>(a[2]), I(a[3]), I(a[4]),
                               I(a[5]), F(a[6]), P<float*>(a[7]), P<float*>(a[8]), s);
    case kHoaV2b:
      return ocrf_hoa_v2b_forward(P<const float*>(a[0]), P<const float*>(a[1]), P<const float*>(a[2]), I(a[3]), I(a[4]),
                                  I(a[5]), P<void*>(a[6]), Z(a[7]), P<float*>(a[8]), s);
    case kHoaStats:
      return ocrf_hoa_channel_stats(P<const float*>(a[0]), I(a[1]), I(a[2]), I(a[3]), I(a[4]), P<float*>(a[5]), s);
    case kHoaMaskGate:
      return ocrf_hoa_opacity_mask_gate(P<const float*>(a[0]), P<const float*>(a[1]), P<const float*>(a[2]),
                                        P<const float*>(a[3]), I(a[4]), I(a[5]), I(a[6]), I(a[7]), I(a[8]),
                                        P<float*>(a[9]), P<float*>(a[10]), s);
    case kStreamWrite32:
      return ocrf_stream_write_value32(P<int*>(a[0]), I(a[1]), s);
    case kRasterPlanBuild:
      return ocrf_raster_plan_build(I(a[0]), I(a[1]), I(a[2]), I(a[3]), P<const float*>(a[4]), P<const float*>(a[5]),
                                    F(a[6]), L(a[7]), P<void*>(a[8]), Z(a[9]), P<void*>(a[10]), Z(a[11]), s);
    // the per-sample step (nothing calibration- or pose-dependent cached): index preparation, pooling on device-side
    // counts, the per-call render.  Host-pointer arguments (grid bounds, pc_range) are read at replay like at the call:
    // the caller keeps them alive (ocrfdet_amd.index_prep._host_floats)
    case kRasterizeForward:
      return ocrf_rasterize_forward(I(a[0]), I(a[1]), I(a[2]), I(a[3]), P<const float*>(a[4]), P<const float*>(a[5]),
                                    P<const float*>(a[6]), P<const float*>(a[7]), F(a[8]), P<const float*>(a[9]),
                                    P<const float*>(a[10]), P<const float*>(a[11]), P<const float*>(a[12]), I(a[13]),
                                    P<float*>(a[14]), P<float*>(a[15]), P<float*>(a[16]), P<uint32_t*>(a[17]),
                                    P<int*>(a[18]), P<uint32_t*>(a[19]), P<int*>(a[20]), P<void*>(a[21]), Z(a[22]), s);
    case kRasterizeForwardSets:
      return ocrf_rasterize_forward_sets(I(a[0]), I(a[1]), I(a[2]), I(a[3]), I(a[4]), P<const float*>(a[5]),
                                         P<const float*>(a[6]), P<const float*>(a[7]), P<const float*>(a[8]), F(a[9]),
                                         P<const float*>(a[10]), P<const float*>(a[11]), P<const float*>(a[12]),
                                         P<const float*>(a[13]), I(a[14]), P<float*>(a[15]), P<float*>(a[16]),
                                         P<float*>(a[17]), P<uint32_t*>(a[18]), P<int*>(a[19]), P<uint32_t*>(a[20]),
                                         P<int*>(a[21]), P<void*>(a[22]), Z(a[23]), s);
    case kPoolDyn:
      return ocrf_bev_pool_v2_nchw_dyn(I(a[0]), I(a[1]), I(a[2]), P<const int*>(a[3]), P<const float*>(a[4]),
                                       P<const float*>(a[5]), P<const int*>(a[6]), P<const int*>(a[7]), P<const int*>(a[8]),
                                       P<const int*>(a[9]), P<const int*>(a[10]), P<float*>(a[11]), I(a[12]), I(a[13]),
                                       I(a[14]), I(a[15]), I(a[16]), P<void*>(a[17]), Z(a[18]), s);
    case kLssPrepare:
      return ocrf_lss_prepare(I(a[0]), I(a[1]), I(a[2]), I(a[3]), I(a[4]), P<const float*>(a[5]), P<const float*>(a[6]),
                              P<const float*>(a[7]), P<const float*>(a[8]), I(a[9]), I(a[10]), I(a[11]), P<int*>(a[12]),
                              P<int*>(a[13]), P<int*>(a[14]), P<int*>(a[15]), P<int*>(a[16]), P<int*>(a[17]),
                              P<void*>(a[18]), Z(a[19]), s);
    case kHtPrepare:
      return ocrf_ht_prepare(I(a[0]), I(a[1]), I(a[2]), I(a[3]), I(a[4]), I(a[5]), I(a[6]), P<const float*>(a[7]),
                             P<const float*>(a[8]), P<const float*>(a[9]), F(a[10]), F(a[11]), F(a[12]), F(a[13]),
                             P<int*>(a[14]), P<int*>(a[15]), P<int*>(a[16]), P<int*>(a[17]), P<int*>(a[18]),
                             P<int*>(a[19]), P<void*>(a[20]), Z(a[21]), s);
    case kGeometryBlocks:
      return ocrf_geometry_blocks(I(a[0]), I(a[1]), P<const float*>(a[2]), P<const float*>(a[3]), P<const float*>(a[4]),
                                  P<const float*>(a[5]), P<const float*>(a[6]), P<const float*>(a[7]),
                                  P<const float*>(a[8]), I(a[9]), I(a[10]), F(a[11]), F(a[12]), P<float*>(a[13]),
                                  P<float*>(a[14]), P<float*>(a[15]), s);
    default:
      return (int)hipErrorInvalidValue;
  }
}

}  // namespace

struct ocrf_step {
  std::vector<Cmd> cmds;
  hipEvent_t events[2 * kMaxStreams * kMaxStreams];      // one per (fork | join, from, to), created on first use
  int device;
};

extern "C" {

// id of an entry point a step can hold, or -1
int ocrf_step_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < kFnCount; ++i)
    if (std::strcmp(name, kFns[i].name) == 0) return i;
  return -1;
}
// number of arguments (without the trailing stream) the entry point takes, or -1
int ocrf_step_fn_args(int fn) { return (fn >= 0 && fn < kFnCount) ? kFns[fn].n_args : -1; }

int ocrf_step_create(ocrf_step** out) {
  if (!out) return (int)hipErrorInvalidValue;
  ocrf_step* s = new ocrf_step();
  for (auto& e : s->events) e = nullptr;
  s->device = -1;
  *out = s;
  return 0;
}

void ocrf_step_destroy(ocrf_step* s) {
  if (!s) return;
  for (auto& e : s->events)
    if (e) (void)hipEventDestroy(e);
  delete s;
}

// a call of entry point `fn` on stream slot `slot` with `n_args` argument words (the stream argument is not among them)
int ocrf_step_add_call(ocrf_step* s, int fn, int slot, int n_args, const uint64_t* args) {
  if (!s || fn < 0 || fn >= kFnCount || slot < 0 || slot >= kMaxStreams || n_args != kFns[fn].n_args || !args)
    return (int)hipErrorInvalidValue;
  Cmd c;
  std::memset(&c, 0, sizeof(c));
  c.kind = 0; c.fn = fn; c.slot = slot;
  std::memcpy(c.a, args, sizeof(uint64_t) * (size_t)n_args);
  s->cmds.push_back(c);
  return 0;
}

// stream `to` continues only after everything issued to `from` so far (kind 1: a fork of `to` off `from`; kind 2: `to`
// joins `from`) — the same event record + wait either way; the two kinds keep their own events
static int add_edge(ocrf_step* s, int kind, int from, int to) {
  if (!s || from < 0 || from >= kMaxStreams || to < 0 || to >= kMaxStreams || from == to) return (int)hipErrorInvalidValue;
  Cmd c;
  std::memset(&c, 0, sizeof(c));
  c.kind = kind; c.from = from; c.to = to;
  s->cmds.push_back(c);
  return 0;
}
int ocrf_step_add_fork(ocrf_step* s, int from, int to) { return add_edge(s, 1, from, to); }
int ocrf_step_add_join(ocrf_step* s, int from, int to) { return add_edge(s, 2, from, to); }

int ocrf_step_size(const ocrf_step* s) { return s ? (int)s->cmds.size() : -1; }

// Replay: every call on the stream of its slot, in recorded order.
int ocrf_step_run(ocrf_step* s, const ocrf_stream_t* streams, int n_streams) {
  if (!s || !streams || n_streams <= 0 || n_streams > kMaxStreams) return (int)hipErrorInvalidValue;
  for (const Cmd& c : s->cmds) {
    if (c.kind == 0) {
      if (c.slot >= n_streams) return (int)hipErrorInvalidValue;
      const int e = call(c, streams[c.slot]);
      if (e != 0) return e;
    } else {
      if (c.from >= n_streams || c.to >= n_streams) return (int)hipErrorInvalidValue;
      hipEvent_t& ev = s->events[((c.kind - 1) * kMaxStreams + c.from) * kMaxStreams + c.to];
      if (!ev) {
        const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
      }
      hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(streams[c.from]));
      if (e != hipSuccess) return (int)e;
      e = hipStreamWaitEvent(static_cast<hipStream_t>(streams[c.to]), ev, 0);
      if (e != hipSuccess) return (int)e;
    }
  }
  return 0;
}

// The two-stream form the hot path uses: slot 0 = the caller's stream, slot 1 = the render stream
int ocrf_hotpath_step(ocrf_step* s, ocrf_stream_t main_stream, ocrf_stream_t side_stream) {
  const ocrf_stream_t streams[2] = {main_stream, side_stream};
  return ocrf_step_run(s, streams, 2);
}

}  // extern "C"
