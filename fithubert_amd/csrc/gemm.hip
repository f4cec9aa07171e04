// Generic MFMA GEMM for gfx950 (see gemm.h): the fp32 parity kernel, the per-shape choice of the 16-bit main loop, the C entry
// points.  The device code lives in gemm_loops.cuh; its 16-bit instantiations in gemm_bf16.hip / gemm_f16.hip.
#include "gemm_loops.cuh"

int fh_g_reserve_cus = 0;
int fh_g_last_loop = 2;  // main loop the last 16-bit launch ran (for the per-launch profile)
int fh_g_last_splitk = 1;
int fh_gemm_path_bf16(const GemmParams& p, int path, hipStream_t stream);  // gemm_bf16.hip
int fh_gemm_path_f16(const GemmParams& p, int path, hipStream_t stream);   // gemm_f16.hip

namespace {

// ------------------------------------------------------------------ 16-bit main-loop selection
// g_gemm_path: 0 = 16x16x32 register staging, 2 = 32x32x16 register staging, 3 = 256x256 ring
// (BK 64, 2 stages), 4 = 256x128 ring (BK 32, 3 stages, 2 workgroups / CU), 5 = 256x256 ring (BK 32, 4 stages),
// 6 = 256x256 ping-pong ring (BK 32, two barrier-staggered wave groups), 7 = 256x256 8-phase loop (BK 64), 9 = per-shape choice
// (default): a table filled by fh_gemm_tune(1) measurements, path 2 for shapes never measured.
// FH_GEMM_PATH = reg | 2..7 forces one loop (A/B comparisons; tools/gemm_bench.py).
int g_gemm_path = -1;
int g_tune_mode = 0;
struct TuneChoice { int path, splitk; };
std::unordered_map<unsigned long long, TuneChoice> g_tune;
// Per-stream set of main loops (fh_gemm_stream_loops): bit i = loop i may be chosen for launches on that stream; 0 / absent = all.
// Every loop of 256 x 256 tiles claims 128 - 160 KiB of a CU's LDS -- one workgroup per CU, nothing of another stream can share it --
// while loops 2 (40 KiB) and 4 (72 KiB) leave room: a stream restricted to them trades some stand-alone speed for co-residence with
// the other streams' kernels (the frozen teacher beside the student).
std::unordered_map<hipStream_t, unsigned> g_stream_mask;
unsigned stream_mask(hipStream_t st) {
  if (g_stream_mask.empty()) return 0u;
  auto it = g_stream_mask.find(st);
  return it == g_stream_mask.end() ? 0u : it->second;
}

void resolve_path() {
  if (g_gemm_path < 0) {
    const char* e = getenv("FH_GEMM_PATH");
    g_gemm_path = !e ? 9 : ((e[0] == 'r' ? 0 : ((e[0] >= '1' && e[0] <= '8') ? e[0] - '0' : 9)));
  }
}
int launch_path(const GemmParams& p, int path, int dtype, hipStream_t stream) {
  return dtype == 2 ? fh_gemm_path_f16(p, path, stream) : fh_gemm_path_bf16(p, path, stream);
}


// shape class of a launch: exact in the small dimensions, quarter-octave buckets in the long one(s) (row counts that follow
// the batch / utterance length), so that length-bucketed batches share their measurements
int qoct(long v) {
  if (v <= 4096) return (int)v;
  int lg = 0;
  while ((v >> (lg + 1)) > 0) ++lg;
  return 100000 + lg * 4 + (int)(((v - (1L << lg)) * 4) >> lg);
}
unsigned long long tune_key_q(const GemmParams& p, int dtype, unsigned mask, int qm, int qk) {
  unsigned long long h = 1469598103934665603ULL;
  auto mix = [&](long v) { h = (h ^ (unsigned long long)v) * 1099511628211ULL; };
  mix(qm); mix(qoct(p.N)); mix(qk); mix(p.X.nseg); mix(p.X.layout); mix(p.Y.layout); mix(p.groups);
  mix(p.ep.atomic); mix(p.ep.atomic ? 0 : p.splitk);
  mix(p.ep.preact != nullptr); mix(p.ep.act); mix(p.ep.gelu_aux != nullptr); mix(p.ep.residual != nullptr);
  if (p.ep.aux_factor) mix(104729);
  // epilogue options that decide whether the streaming loop 8 can take the launch at all (stream_ok): a class measured without them
  // must not hand "loop 8" to a launch that has them -- the padded-batch projection would drop to the 128-tile fallback
  if (p.ep.row_len || p.ep.out_f32 || p.ep.tail_cols || (p.c_rpb > 0 && p.c_bstride != (long)p.c_rpb * p.ldc))
    mix(2750159 + (p.ep.row_len ? 1 : 0) + (p.ep.out_f32 ? 2 : 0) + (p.ep.tail_cols ? 4 : 0) +
        ((p.c_rpb > 0 && p.c_bstride != (long)p.c_rpb * p.ldc) ? 8 : 0));
  if (p.Y.seg1_limit > 0) mix(7919 + p.Y.seg1_limit);
  if (dtype == 2) mix(15485863);  // f16 launches keep their own measurements
  if (mask) mix(0x5bd1e995L + mask);  // ... and so do launches of a stream with a restricted set of loops
  return h;
}
unsigned long long tune_key(const GemmParams& p, int dtype, unsigned mask = 0u) {
  return tune_key_q(p, dtype, mask, qoct(p.M), qoct(p.Kseg));
}
// A launch whose own shape class was never measured takes the choice of the nearest measured class that differs only in the long
// dimension(s) (rows for the forward / data-gradient launches, the reduction for the weight gradients): StudentEngine.autotune keys
// its passes on the utterance length's quarter octave, while every conv layer's row count crosses ITS quarter-octave boundaries at
// other lengths -- a length-bucketed epoch (445 bucket shapes in train-clean-100) would otherwise drop such launches to the
// 128-tile fallback (B = 64 x 15.4 s: 51 ms instead of 41).  A loop that cannot take the launch falls through to loop 2 in launch_path.
const TuneChoice* tune_neighbour(const GemmParams& p, int dtype, unsigned mask) {
  const int qm = qoct(p.M), qk = qoct(p.Kseg);
  const bool vm = p.M > 4096, vk = p.Kseg > 4096;
  if (!vm && !vk) return nullptr;
  for (int d = 1; d <= 12; ++d)   // three octaves either way, nearest first
    for (int dm = vm ? -d : 0; dm <= (vm ? d : 0); ++dm)
      for (int dk = vk ? -d : 0; dk <= (vk ? d : 0); ++dk) {
        if ((dm < 0 ? -dm : dm) != d && (dk < 0 ? -dk : dk) != d) continue;
        auto it = g_tune.find(tune_key_q(p, dtype, mask, qm + dm, qk + dk));
        if (it != g_tune.end()) return &it->second;
      }
  return nullptr;
}
int ring_splitk(const GemmParams& p, int tm, int tn, int target) {
  const long tiles = (long)((p.M + tm - 1) / tm) * ((p.N + tn - 1) / tn) * p.groups;
  const long red = (long)p.Kseg * p.X.nseg;
  long sk = (target + tiles - 1) / tiles;
  const long maxsk = red / 512 > 0 ? red / 512 : 1;
  if (sk > maxsk) sk = maxsk;
  if (sk < 1) sk = 1;
  if (sk > 256) sk = 256;
  return (int)sk;
}
// Times the candidate loops for this launch's shape class on `stream` (blocking; the output is written / accumulated
// several times, so only throw-away passes -- StudentEngine.autotune -- may run with tuning on).
TuneChoice tune_measure(const GemmParams& p, int dtype, hipStream_t stream) {
  std::vector<TuneChoice> cand;
  const unsigned mask = stream_mask(stream);
  cand.push_back({2, p.splitk});   // (loop 2 takes any launch: it stays the fallback of a restricted stream too)
  auto add = [&](int path, int sk) {
    if (mask && !((mask >> path) & 1u)) return;
    for (auto& c : cand)
      if (c.path == path && c.splitk == sk) return;
    cand.push_back({path, sk});
  };
  const long red = (long)p.Kseg * p.X.nseg;
  if (red >= 128) {   // (from two 64-deep K tiles on: the K = 128 conv layer is worth a look for the lean bf16 epilogue of loop 7)
    if (p.ep.atomic) {
      const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
      // the atomic epilogue costs ~9 us per extra split of a 12 x 480 x 480 gradient (tools/gemm_wgrad_sweep.py): fewer workgroups
      // than CUs can win, so the candidates start below one wave of workgroups
      // both roundings of target / tiles: rounding up alone puts e.g. 3 tiles x 86 splits = 258 workgroups on 256 CUs (two
      // rounds for the last two workgroups); 3 x 85 fills one round
      for (int target : {64, 128, 192, 256, 512}) {
        const int sk_up = ring_splitk(p, 256, 256, target);
        long sk_dn = target / (tiles256 * p.groups);
        if (sk_dn > sk_up) sk_dn = sk_up;   // ring_splitk's caps (reduction length, 256)
        for (int sk : {sk_up, (int)(sk_dn < 1 ? 1 : sk_dn)}) {
          for (int remap : {0, FH_SPLIT_REMAP}) {
            if (remap && (sk < 2 || tiles256 < 2)) continue;  // nothing to co-locate
            if (ring_ok(p, 64)) { add(3, sk | remap); add(7, sk | remap); }
            if (ring_ok(p, 32)) add(5, sk | remap);
          }
        }
      }
      add(2, ring_splitk(p, 128, 128, 1024));
    } else {
      add(0, p.splitk);
      if (small_ok(p) && (long)p.M * p.N * p.groups <= (long)8192 * 2048) add(1, p.splitk);  // few tiles: the one-trip small-M loop
      if (ring_ok(p, 64)) add(3, p.splitk);
      if (ring_ok(p, 32)) { add(4, p.splitk); add(5, p.splitk); add(6, p.splitk); }
      if (ring_ok(p, 64)) { add(7, p.splitk); if (stream_ok(p)) add(8, p.splitk); }
    }
  }
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  TuneChoice best = cand[0];
  float best_ms = 1e30f, ms8 = 1e30f;
  for (auto& c : cand) {
    GemmParams q = p;
    q.splitk = c.splitk;
    if (launch_path(q, c.path, dtype, stream) != 0) continue;
    float ms = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {  // best of two 3-launch timings: the candidates are often within a few percent
      hipEventRecord(a, stream);
      for (int i = 0; i < 3; ++i) launch_path(q, c.path, dtype, stream);
      hipEventRecord(b, stream);
      if (hipEventSynchronize(b) != hipSuccess) break;
      float t = 0.f;
      hipEventElapsedTime(&t, a, b);
      ms = t < ms ? t : ms;
    }
    if (ms < best_ms) { best_ms = ms; best = c; }
    if (c.path == 8) ms8 = ms;
  }
  // loops 7 and 8 share their main loop and are within a few percent of each other when timed alone, so timing noise decided
  // between them.  Loop 8 takes the launch unless loop 7 wins by more than FH_GEMM_PREFER8 percent (default 4): in the step these
  // launches run beside another stream's kernels, and the persistent form keeps its CUs and its XCD-local tile order for the whole
  // launch (its launches stay near their stand-alone duration; loop 7's workgroups interleave with the other stream's and stretch
  // 2 - 4 x, which packs the step 0.5 - 1 % tighter but leaves no per-kernel figure one can read: DESIGN.md section 5)
  static int prefer8 = -1;
  if (prefer8 < 0) {
    const char* e = getenv("FH_GEMM_PREFER8");
    prefer8 = e ? atoi(e) : 4;
  }
  if (best.path == 7 && ms8 < best_ms * (1.f + 0.01f * prefer8)) best = {8, p.splitk};
  hipEventDestroy(a);
  hipEventDestroy(b);
  return best;
}

int launch_16(const GemmParams& p, int dtype, hipStream_t stream) {
  resolve_path();
  if (p.ep.c0_wpart) return launch_path(p, 4, dtype, stream);   // the conv-0 weight-gradient epilogue lives in loop 4
  if (p.ep.kd_teacher) {   // the KD-loss epilogue is compiled into loops 7 and 4 (FH_KD_LOOP picks; default 4: two workgroups per CU, so one
    static int kd_loop = -1;   // workgroup's operand-bound epilogue runs under the other's K loop)
    if (kd_loop < 0) {
      const char* e = getenv("FH_KD_LOOP");
      kd_loop = (e && e[0] == '7') ? 7 : 4;
    }
    return launch_path(p, kd_loop, dtype, stream);
  }
  if (g_gemm_path != 9) return launch_path(p, g_gemm_path, dtype, stream);
  const unsigned long long key = tune_key(p, dtype, stream_mask(stream));
  auto it = g_tune.find(key);
  if (it == g_tune.end()) {
    if (!g_tune_mode) {
      const TuneChoice* nb = tune_neighbour(p, dtype, stream_mask(stream));
      // no measurement for this class or a neighbouring one (a caller that never switched the tuner on): launches with many rows
      // take the 256-tile 8-phase loops -- streaming where the epilogue allows it (launch_path_h falls back from 8 to 7 to the
      // 128-tile loop by itself); at M = 63,744, N = K = 480 they run 40 us against the 128-tile loop's 59 -- the rest loop 2
      // (the tuner's own gate on the reduction depth applies here too: it never offers loops 3 - 8 below 128, i.e. a single K-tile)
      if (!nb) {
        const bool deep = (long)p.Kseg * p.X.nseg >= 128;
        return launch_path(p, (!p.ep.atomic && p.M >= 4096 && p.N >= 128 && deep) ? 8 : 2, dtype, stream);
      }
      TuneChoice c = *nb;
      if (p.ep.atomic) {   // a split count measured on a longer reduction: keep every split at least 512 deep, as the candidates are
        const long red = (long)p.Kseg * p.X.nseg;
        const int cap = (int)(red / 512 > 0 ? red / 512 : 1), sk = c.splitk & (FH_SPLIT_REMAP - 1);
        if (sk > cap) c.splitk = cap | (c.splitk & FH_SPLIT_REMAP);
      }
      it = g_tune.emplace(key, c).first;   // (remembered: the search runs once per class)
    } else {
      it = g_tune.emplace(key, tune_measure(p, dtype, stream)).first;
    }
  }
  GemmParams q = p;
  if (p.ep.atomic) q.splitk = it->second.splitk;
  return launch_path(q, it->second.path, dtype, stream);
}

}  // namespace

// ------------------------------------------------------------------ optional per-launch HIP-event timing (bench.py roofline)
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int variant; int M, N, K, groups, splitk, tag, atomic, loop, xl, yl; };
int g_prof_tag = 0;
std::vector<ProfRec> g_prof;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
bool g_prof_on = false;
}  // namespace

extern "C" int fh_gemm_last_loop(void) { return fh_g_last_loop; }
extern "C" int fh_gemm_reserve_cus(int n) {
  const int old = fh_g_reserve_cus;
  if (n >= 0) fh_g_reserve_cus = n;
  return old;
}
namespace { std::unordered_map<hipStream_t, int> g_stream_reserve; }
int fh_stream_reserve_cus(hipStream_t st) {
  if (g_stream_reserve.empty()) return 0;
  auto it = g_stream_reserve.find(st);
  return it == g_stream_reserve.end() ? 0 : it->second;
}
extern "C" int fh_gemm_stream_reserve_cus(fh_stream_t stream, int n) {
  if (n > 0) g_stream_reserve[(hipStream_t)stream] = n; else g_stream_reserve.erase((hipStream_t)stream);
  return 0;
}
// A HIP stream whose kernels may only run on `count` CUs (hipExtStreamCreateWithCUMask): mask bits [first, first + count) of the
// device's CU mask.  The KFD deals consecutive mask bits round-robin over the XCDs, so a contiguous bit range is an (almost) equal
// share of every XCD (tools/cu_partition.py prints the census of a masked launch).  Loop 8's persistent grid on that stream is
// sized to the share (fh_gemm_stream_reserve_cus).
extern "C" int fh_stream_create_cu_range(int first, int count, fh_stream_t* out) {
  if (!out || first < 0 || count < 1) return -1;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -121;
  const int ncu = prop.multiProcessorCount;
  if (first + count > ncu) return -2;
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) return -3;
  *out = (fh_stream_t)st;
  if (count < ncu) g_stream_reserve[st] = ncu - (count & ~7);
  return 0;
}
// Where a launch on `stream` lands: one 64-thread workgroup per entry records (XCC id << 16 | HW_ID[15:0]) after spinning ~20 us so
// that the workgroups spread over every CU the stream may use.
namespace {
__global__ void cu_census_kernel(unsigned* out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf;    // HW_REG_XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) & 0xffff;  // HW_REG_HW_ID: wave / simd / pipe / cu / sh / se
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}                               // 100 MHz ticks
  if (threadIdx.x == 0) out[blockIdx.x] = xcc << 16 | hw;
}
}  // namespace
extern "C" int fh_stream_cu_census(unsigned* out, int n, fh_stream_t stream) {
  if (!out || n < 1) return -1;
  hipLaunchKernelGGL(cu_census_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int fh_stream_destroy(fh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!st) return 0;
  g_stream_reserve.erase(st);
  g_stream_mask.erase(st);
  return hipStreamDestroy(st) == hipSuccess ? 0 : -1;
}
extern "C" int fh_gemm_stream_loops(fh_stream_t stream, unsigned mask) {
  if (mask) g_stream_mask[(hipStream_t)stream] = mask; else g_stream_mask.erase((hipStream_t)stream);
  return 0;
}
extern "C" int fh_gemm_set_path(int path) {
  g_gemm_path = path;
  return 0;
}
bool fh_gemm_tuning() { return g_tune_mode != 0; }  // launches are being timed: callers keep everything on one stream
extern "C" int fh_gemm_tune(int mode) {
  if (mode < 0) g_tune.clear();
  g_tune_mode = mode > 0;
  return (int)g_tune.size();
}
// table exchange (so that a profiled run can reuse the choices of an earlier run instead of re-measuring)
extern "C" int fh_gemm_tune_export(unsigned long long* keys, int* paths, int* splitks, int cap) {
  int n = 0;
  for (auto& kv : g_tune) {
    if (n < cap && keys && paths && splitks) {
      keys[n] = kv.first;
      paths[n] = kv.second.path;
      splitks[n] = kv.second.splitk;
    }
    ++n;
  }
  return n;
}
extern "C" int fh_gemm_tune_import(const unsigned long long* keys, const int* paths, const int* splitks, int n) {
  for (int i = 0; i < n; ++i) g_tune[keys[i]] = TuneChoice{paths[i], splitks[i]};
  return (int)g_tune.size();
}
// caller-chosen tag attached to the launches recorded from now on (bench.py: 1 while the frozen teacher's forward is queued)
extern "C" int fh_prof_tag(int tag) {
  const int old = g_prof_tag;
  g_prof_tag = tag;
  return old;
}
// per-launch form of fh_prof_collect: rows of {M, N, K (whole reduction), groups, X layout, Y layout, loop, flags (bit 0 atomic output,
// bit 1 GELU' operand = a conv / fc1 data gradient, bit 2 conv-0 weight-gradient epilogue), tag} and the
// launch's duration / algorithmic FLOPs; returns the number of launches recorded (rows beyond cap are dropped) and clears the log
extern "C" int fh_prof_collect_launches(double* ms, double* flops, int* rows /*[cap][9]*/, int cap) {
  int n = 0;
  const char* trace = getenv("FH_GEMM_TRACE");   // the same per-launch lines fh_prof_collect writes
  FILE* tf = trace && *trace ? fopen(trace, "a") : nullptr;
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && n < cap) {
      if (tf) fprintf(tf, "%d %d %d %d %d %d %d %d %.4f %d\n", r.M, r.N, r.K, r.groups, r.xl, r.yl, r.loop, r.splitk, t, r.atomic);   // (last: the flag bits)
      ms[n] = t;
      flops[n] = r.flops;
      int* q = rows + 9 * n;
      q[0] = r.M; q[1] = r.N; q[2] = r.K; q[3] = r.groups; q[4] = r.xl; q[5] = r.yl; q[6] = r.loop; q[7] = r.atomic; q[8] = r.tag;
      ++n;
    }
    g_pool.emplace_back(r.a, r.b);
  }
  if (tf) fclose(tf);
  g_prof.clear();
  return n;
}
extern "C" int fh_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
// sums per slot = loop * 8 + (dtype*4 + X layout*2 + Y layout), loop = the main loop that ran (fh_gemm_set_path numbering,
// 0 for fp32); returns the number of launches recorded and clears the log
extern "C" int fh_prof_collect(double* ms /*[64]*/, double* flops /*[64]*/, long* launches /*[64]*/) {
  for (int i = 0; i < 64; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  int n = 0;
  // FH_GEMM_TRACE=<file>: one line per launch (shape, operand layouts, loop, split, ms) -- a tuning aid
  const char* trace = getenv("FH_GEMM_TRACE");
  FILE* tf = trace && *trace ? fopen(trace, "a") : nullptr;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    if (tf)
      fprintf(tf, "%d %d %d %d %d %d %d %d %.4f %d\n", r.M, r.N, r.K, r.groups, (r.variant >> 1) & 1, r.variant & 1,
              (r.variant >= 8 && r.variant < 12) ? 8 : r.variant >> 3, r.splitk, t, r.atomic);
    ms[r.variant] += t;
    flops[r.variant] += r.flops;
    launches[r.variant] += 1;
    g_pool.emplace_back(r.a, r.b);
    ++n;
  }
  if (tf) fclose(tf);
  g_prof.clear();
  return n;
}

int fh_gemm_launch(const GemmParams& p, int dtype, hipStream_t stream) {
  if (dtype < 0 || dtype > 2) return -100;
  const int vec = dtype == 0 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0 || p.Kseg <= 0) return 0;
  // vector-access preconditions (all hold for the student's shapes; fail loudly otherwise)
  if (p.N % 4 != 0 || (!p.ep.atomic && p.N % 8 != 0)) return -101;
  if (p.X.layout == FH_R_CONTIG && (p.Kseg % vec || p.X.ld % vec || p.X.bstride % vec)) return -102;
  if (p.Y.layout == FH_R_CONTIG && (p.Kseg % vec || p.Y.ld % vec || p.Y.bstride % vec)) return -103;
  if (p.X.layout == FH_R_STRIDED && (p.M % vec || p.X.ld % vec || p.X.bstride % vec)) return -104;
  if (p.Y.layout == FH_R_STRIDED && (p.N % vec || p.Y.ld % vec || p.Y.bstride % vec)) return -105;
  if (p.X.nseg != p.Y.nseg || p.X.nseg < 1 || p.X.nseg > 2) return -106;
  if ((p.splitk & 0xffff) > 1 && !p.ep.atomic) return -107;
  if (!p.ep.atomic && (p.ldc % 8 || p.c_bstride % 8 || p.c_gstride % 8)) return -108;
  if (p.ep.row_len && p.c_rpb <= 0) return -109;
  if ((p.X.segoff[0] | p.X.segoff[1] | p.Y.segoff[0] | p.Y.segoff[1] | p.X.gstride | p.Y.gstride) % vec) return -110;
  if (p.X.seg1_limit % 8 || p.Y.seg1_limit % 8 || p.ep.tail_cols % 8 || (p.ep.tail_cols > 0 && (p.c_rpb <= 0 || p.ep.atomic))) return -111;
  if (p.ep.c0_wpart) {     // conv-0 weight-gradient epilogue: one utterance per group, 128 channels, 16-bit
    if (dtype == 0 || p.N != 128 || p.ep.c0_K < 1 || p.ep.c0_K > 10 || p.ep.c0_stride < 1 || 256 * p.ep.c0_stride + 16 > 2048 || p.ep.atomic ||
        p.ep.residual || p.ep.gelu_aux || p.ep.row_len || p.ep.kd_teacher || (p.splitk & 0xffff) > 1 || p.X.layout != FH_R_CONTIG ||
        p.Y.layout != FH_R_STRIDED || !p.ep.c0_wave || !p.ep.c0_w || !p.ep.c0_gamma || !p.ep.c0_beta || !p.ep.c0_mean || !p.ep.c0_rstd)
      return -114;
  }
  if (p.ep.kd_teacher) {   // KD-loss epilogue: the forms that implement it
    const long wrap0 = p.c_bstride - (long)p.c_rpb * p.ldc;
    if (p.c_rpb <= 0 || wrap0 != 0 || p.ep.kd_P <= 0 || p.ep.kd_P % 8 || p.N % p.ep.kd_P || p.ep.atomic || p.ep.residual || p.ep.gelu_aux ||
        p.ep.row_len || p.ep.out_f32 || p.ep.preact || p.ep.tail_cols || (p.splitk & 0xffff) > 1 || !p.ep.kd_layer || !p.ep.kd_w ||
        !p.ep.kd_partials || p.ep.kd_tdtype < 0 || p.ep.kd_tdtype > 2 || p.ep.act != FH_ACT_NONE || p.ep.drop.p > 0.f)
      return -112;
  }
  if (!g_prof_on) return dtype == 0 ? launch_t<float>(p, stream) : launch_16(p, dtype, stream);
  ProfRec r;
  if (!g_pool.empty()) {
    r.a = g_pool.back().first; r.b = g_pool.back().second;
    g_pool.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -120;
  }
  // reduction segments per output column: columns at or beyond Y.seg1_limit run one segment (the single-product k = 3, s = 2
  // conv data gradient), so its work is Kseg * (N + min(N, seg1_limit)), not 2 * Kseg * N
  const double ncols = (p.X.nseg == 2 && p.Y.seg1_limit > 0) ? (double)p.N + (p.N < p.Y.seg1_limit ? p.N : p.Y.seg1_limit)
                                                             : (double)p.N * p.X.nseg;
  r.flops = 2.0 * p.M * ncols * (double)p.Kseg * p.groups;
  r.variant = (dtype ? 4 : 0) + p.X.layout * 2 + p.Y.layout;
  r.M = p.M; r.N = p.N; r.K = p.Kseg * p.X.nseg; r.groups = p.groups; r.splitk = p.splitk;
  r.tag = g_prof_tag; r.atomic = (p.ep.atomic ? 1 : 0) | (p.ep.gelu_aux ? 2 : 0) | (p.ep.c0_wpart ? 4 : 0);  // bit 0 atomic output, bit 1 GELU' operand, bit 2 conv-0 epilogue
  r.xl = p.X.layout; r.yl = p.Y.layout; r.loop = 0;
  hipEventRecord(r.a, stream);
  const int rc = dtype == 0 ? launch_t<float>(p, stream) : launch_16(p, dtype, stream);
  if (dtype != 0) {
    // slot = loop * 8 + 4 + layouts; the persistent 8-phase loop (8) takes the otherwise unused slots 8 + layouts
    r.variant = fh_g_last_loop == 8 ? 8 + p.X.layout * 2 + p.Y.layout : r.variant + 8 * (fh_g_last_loop & 7);
    r.splitk = fh_g_last_splitk;
    r.loop = fh_g_last_loop;
  }
  hipEventRecord(r.b, stream);
  g_prof.push_back(r);
  return rc;
}
