// Paged GQA decode attention for gfx950 (replaces FlashInfer BatchDecodeWithPagedKVCache behind
// paged_attention_decode_cuda / paged_attention_decode_split_kv_cuda, reference
// csrc/paged_attention.cu:77-230).
//
// HBM-bound KV scan.  Grid = (plan slots, kv_heads); a workgroup of 4 waves owns one KV head of one
// (request, KV chunk) and all GROUP query heads that share it, so every K/V byte is read once per
// group, not once per query head.  Inside a wave a K row (head_dim bf16) is spread over
// LPT = head_dim/8 lanes with one 16-byte load each -> a single load instruction fetches
// 64/LPT complete rows, fully coalesced (the page-first NHD cache keeps a head's row contiguous).
// q.k uses v_dot2c_f32_bf16 on the packed bf16 pairs and a DPP butterfly inside the 16-lane row
// (no LDS, no bpermute); each lane row keeps its own online-softmax state (m, l, o[8 dims]) over the
// tokens it sees, so the main loop has no cross-row traffic at all; the 4 x (64/LPT) partial states
// of the workgroup are merged once through LDS.  exp2 with sm_scale*log2(e) folded in, fp32 state.
//
// Partition-KV ("split-K", mandatory on a 256-CU part: bs=1 exposes only kv_heads=8 workgroups
// otherwise) writes a normalised bf16 partial + fp32 log2-sum-exp per (slot, q head) into the
// caller's tmp_v / tmp_s exactly like FlashInfer, and merge_states_kernel combines slots
// o_indptr[b]..o_indptr[b+1].  kv_len always comes from the page table
// ((pages-1)*page_size + last_page_len), kv_chunk_size_ptr[0] is read only when partitioning.
#include "attn_decode_core.h"
#include "pegainfer_kernels_ext.h"

namespace pk {

template <bool PARTITION>
__device__ __forceinline__ ChunkInfo decode_chunk(const DecodeAttnArgs& a, int slot) {
  ChunkInfo c;
  c.b = a.request_indices ? a.request_indices[slot] : slot;
  c.pbase = a.page_indptr[c.b];
  const int npages = a.page_indptr[c.b + 1] - c.pbase;
  c.kv_len = npages > 0 ? (npages - 1) * a.page_size + a.last_page_len[c.b] : 0;
  c.lo = 0;
  c.hi = c.kv_len;
  if (PARTITION) {
    const int chunk = a.kv_chunk_size_ptr[0];
    c.lo = a.kv_tile_indices[slot] * chunk;
    c.hi = c.lo + chunk < c.kv_len ? c.lo + chunk : c.kv_len;
    if (c.lo > c.hi) c.lo = c.hi;
  }
  return c;
}

// The KV scan + in-workgroup merge, given the (already normalised / rotated) bf16 q fragments (attn_decode_core.h
// holds the arithmetic; this is the real-workgroup driver: NW waves, __syncthreads, static LDS).
template <int D, int GROUP, bool PARTITION, int NW>
__device__ __forceinline__ void decode_attn_body(const DecodeAttnArgs& a, const ChunkInfo& ci, const u32x4 (&qv)[GROUP],
                                                 int slot, int kvh) {
  typedef AttnScan<D, GROUP> Scan;
  constexpr int NPART = NW * Scan::TPI;  // partial softmax states per workgroup (NW waves)
  __shared__ float sm_m[NPART * GROUP];
  __shared__ float sm_l[NPART * GROUP];
  __shared__ __attribute__((aligned(16))) float sm_o[NPART * GROUP * D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Scan st;
  st.init();
  PK_ATTN_STAMP(a, slot, kvh, 2);
  st.scan(a, ci, qv, kvh, lane, wave, NW);
  PK_ATTN_STAMP(a, slot, kvh, 3);
  // merge the workgroup's NPART partial states
  st.store_state(sm_m, sm_l, sm_o, wave * Scan::TPI + lane / Scan::LPT, lane);
  // two-stage merge (round 6, attn_decode_core.h): every wave folds its own TPI lane-row states (no barrier: own slots), then
  // one thread per (head, 8 output dims) merges the NW folds.  PEGAINFER_ATTN_FOLD=0 at build time is not offered: one
  // arithmetic for every form.
  static_assert(64 % (D / 8) == 0, "an element sweep of 64 lanes must cover whole heads");
  for (int e = lane; e < GROUP * (D / 8); e += 64) attn_fold_wave<D, GROUP, Scan::TPI>(sm_m, sm_l, sm_o, wave, e);
  __syncthreads();
  // one thread per (head, 8 output dims): 16-byte stores.  With merge_counters the partials are published
  // write-through (sc1): they are read by a workgroup on another XCD later in this same launch.
  const bool publish = PARTITION && a.merge_counters != nullptr;
  for (int e = threadIdx.x; e < GROUP * (D / 8); e += NW * 64)
    attn_finish_part<D, GROUP, PARTITION>(a, ci.b, slot, kvh, e, NW, sm_m, sm_l, sm_o, publish, false, Scan::TPI);
  if (publish) {
    // "last workgroup done" merge without cache-wide fences (guide: sc1 payload -> vmcnt(0) -> counter; the
    // reader uses sc1 loads): every chunk's partials are write-through, the ticket is a relaxed agent atomic,
    // and the workgroup that draws n-1 merges this head group and re-arms the counter for the next launch.
    __shared__ int sm_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PK_ATTN_STAMP(a, slot, kvh, 4);
    const int b = ci.b, num_kv_heads = a.num_kv_heads, num_qo_heads = a.num_qo_heads;
    const int s0 = a.o_indptr[b], s1 = a.o_indptr[b + 1];
    if (threadIdx.x == 0) {
      int* ctr = a.merge_counters + (size_t)(b * num_kv_heads + kvh) * kMergeCtrStride;
      const int last = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s1 - s0 - 1;
      if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sm_last = last;
    }
    __syncthreads();
    PK_ATTN_STAMP(a, slot, kvh, 5);
    if (sm_last) {
      for (int h = wave; h < GROUP; h += NW) {
        const int head = kvh * GROUP + h;
        if (a.done_ctr)   // the rows are consumed inside this launch (attn_oproj_kernel): write-through
          merge_one<D, true, true>(a.tmp_v, a.tmp_s, s0, s1, head, num_qo_heads,
                                   a.o_out + ((size_t)b * num_qo_heads + head) * D);
        else
          merge_one<D, true>(a.tmp_v, a.tmp_s, s0, s1, head, num_qo_heads,
                             a.o_out + ((size_t)b * num_qo_heads + head) * D);
      }
      if (a.done_ctr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // workgroup-uniform branch (sm_last): every merging wave's rows are acknowledged
        if (threadIdx.x == 0)
          __hip_atomic_fetch_add(a.done_ctr + (size_t)(b * num_kv_heads + kvh) * a.done_stride, 1, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
      PK_ATTN_STAMP(a, slot, kvh, 6);
      if (a.trace && threadIdx.x == 0) a.trace[((size_t)slot * a.num_kv_heads + kvh) * 8 + 7] = 1ull;
    }
  }
}

// ---- kernels: reference-ABI form (q already normalised + rotated, K/V already in the cache) ----
template <int D, int GROUP, bool PARTITION, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(const DecodeAttnArgs a) {
  const int slot = blockIdx.x, kvh = blockIdx.y;
  if (PARTITION && a.block_valid_mask && !a.block_valid_mask[slot]) return;
  const ChunkInfo ci = decode_chunk<PARTITION>(a, slot);
  const int sub = (threadIdx.x & 63) % (D / 8);
  u32x4 qv[GROUP];
#pragma unroll
  for (int h = 0; h < GROUP; ++h)
    qv[h] = *reinterpret_cast<const u32x4*>(a.q + ((size_t)ci.b * a.num_qo_heads + kvh * GROUP + h) * D + sub * 8);
  decode_attn_body<D, GROUP, PARTITION, NW>(a, ci, qv, slot, kvh);
}

// ---- fused form (head_dim 128): per-head q/k RMSNorm + RoPE and the KV append folded into the prologue.
// Every workgroup normalises + rotates its GROUP query heads from the raw qkv row (cheap, redundant across
// the request's chunks); the one workgroup per (request, kv head) whose chunk contains the new position also
// normalises + rotates the new K row, writes K and V into the page (same bytes paged_kv_scatter_cuda would
// write) and only then scans.  Replaces qk_norm_rope + paged_kv_scatter + decode attention: 3 launches -> 1.
template <int GROUP, bool PARTITION, int NW>
__device__ __forceinline__ void fused_decode_attn_workgroup(const DecodeAttnArgs& a, int slot, int kvh) {
  constexpr int D = 128;
  ChunkInfo ci;
  int pos;
  PK_ATTN_STAMP(a, slot, kvh, 0);
  if (a.slot_desc) {
    const u32x4 d0 = *reinterpret_cast<const u32x4*>(a.slot_desc + 8 * slot);
    const u32x2 d1 = *reinterpret_cast<const u32x2*>(a.slot_desc + 8 * slot + 4);
    ci.b = (int)d0.x; ci.lo = (int)d0.y; ci.hi = (int)d0.z; ci.pbase = (int)d0.w;
    pos = (int)d1.x; ci.kv_len = (int)d1.y;
    if (ci.lo < 0) return;
  } else {
    if (PARTITION && a.block_valid_mask && !a.block_valid_mask[slot]) return;
    ci = decode_chunk<PARTITION>(a, slot);
    pos = a.positions[ci.b];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane & 15, grp = lane >> 4;
  const int q_dim = a.num_qo_heads * D, kv_dim = a.num_kv_heads * D;
  const Half* row = a.qkv + (size_t)ci.b * (q_dim + 2 * kv_dim);
  const Half* crow = a.cos_cache + (size_t)pos * D;
  const Half* srow = a.sin_cache + (size_t)pos * D;
  PK_ATTN_STAMP(a, slot, kvh, 1);
  u32x4 qv[GROUP];
#pragma unroll
  for (int h = 0; h < GROUP; ++h) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(row + (size_t)(kvh * GROUP + h) * D + sub * 8);
    qv[h] = head_norm_rope16(x, a.q_norm_w, crow, srow, sub, a.eps);
  }
  const bool owns_new = pos >= ci.lo && pos < ci.hi;  // workgroup-uniform
  if (owns_new) {
    if (wave == 0 && grp == 0) {
      const u32x4 xk = *reinterpret_cast<const u32x4*>(row + q_dim + (size_t)kvh * D + sub * 8);
      const u32x4 kn = head_norm_rope16(xk, a.k_norm_w, crow, srow, sub, a.eps);
      const u32x4 xv = *reinterpret_cast<const u32x4*>(row + q_dim + kv_dim + (size_t)kvh * D + sub * 8);
      const int page = a.page_indices[ci.pbase + pos / a.page_size];
      const long base = (long)page * a.stride_page + ((long)(pos % a.page_size) * a.num_kv_heads + kvh) * D + sub * 8;
      Half* kvw = const_cast<Half*>(a.kv);
      *reinterpret_cast<u32x4*>(kvw + base + a.k_off) = kn;
      *reinterpret_cast<u32x4*>(kvw + base + a.v_off) = xv;
    }
    __syncthreads();  // workgroup-scope release/acquire: the new row is visible to the scanning waves
  }
  decode_attn_body<D, GROUP, PARTITION, NW>(a, ci, qv, slot, kvh);
}


template <int GROUP, bool PARTITION, int NW>
__global__ __launch_bounds__(NW * 64) void fused_decode_attn_kernel(const DecodeAttnArgs a) {
  fused_decode_attn_workgroup<GROUP, PARTITION, NW>(a, blockIdx.x, blockIdx.y);
}

// ---- fused decode attention + o_proj, ONE launch, single-request steps (round 3).
// The attention launch of a bs = 1 step is a chain of dependent round trips (~9-10 us) during which HBM idles, and the
// o_proj GEMV behind it pays a kernel boundary, an x hand-off and a cold weight burst (6.4 us for 21 MB).  Here the
// launch grid is the attention's own slots x kv_heads grid of 8-wave workgroups (one per CU); the workgroups of the
// plan's VALID slots run the fused attention exactly as the stand-alone kernel does, the workgroups of the padding
// slots - idle exits in the stand-alone launch, (slots - valid) x kv_heads of them - take the o_proj: each requests
// its share of the rows into registers (the two wave quads of a workgroup act exactly like KSPLIT = 4 GEMV workgroups:
// K blocks dealt round-robin to the 4 waves, fixed-order LDS combine - the bits of gemv_fused_kernel), waits until the
// merging workgroups have published the attention row write-through and arrived on `done`, takes the row with
// cache-bypassing loads and finishes its dot products from registers.  No attention workgroup ever waits for anything
// (the protocol needs no co-residency guarantee beyond "the attention workgroups get to run"); every wait is bounded.
// Phase history (profiles/r3_attn_oproj_*): every workgroup holding o_proj rows THROUGH its attention slowed the
// attention by 2.4 us (256 registers, weight bursts in front of its dependent loads) and left the merging workgroups,
// which could only request their rows last, as a 1.5 us tail.
struct OprojArgs {
  const Half* W; Half* Y; int M, K;
  int slots;                       // launch slots of the (single) request; how many carry a KV chunk is read from slot_desc
  int done_target;                 // arrivals to wait for: num_kv_heads
  uint32_t* status;
  int hold_ticks;                  // 100 MHz ticks the o_proj workgroups hold their burst back (the attention's first
                                   // dependent loads - slot record, q row, page ids, K / V tile - go first)
  unsigned long long wait_ticks;   // bound of the o_proj workgroups' wait for the attention rows (30 ms; PEGAINFER_OPROJ_WAIT_TICKS
                                   // = 1 makes it expire at once: the test hook of the host's re-run path)
};
constexpr int kOprojMaxRows = 14;  // rows per wave quad (12 until round 5; 14 admits 20 KV chunks = 96 o_proj workgroups for hidden 2560)

template <int GROUP, int NB>   // NB = requests of the step (1, or 2 since round 5: VERDICT r4 item 4a)
__global__ __launch_bounds__(512) void attn_oproj_kernel(const DecodeAttnArgs a, const OprojArgs g) {
  constexpr int NW = 8;
  const int slot = (int)blockIdx.x % g.slots, kvh = (int)blockIdx.x / g.slots;   // the dim3(slots, kv_heads) order
  // debug stamps of the o_proj workgroups (pegainfer_debug_attn_trace): 8 words per workgroup behind the attention's
  // 64 * kv_heads records: entry, rows requested, rows of the attention seen (flag), x staged, exit
  unsigned long long* ot = a.trace ? a.trace + ((size_t)64 * a.num_kv_heads + blockIdx.x) * 8 : nullptr;
  // valid slots of request 0 = its partial count o_indptr[1] - o_indptr[0], words 6 / 7 of slot 0's record (a captured
  // graph replays this launch while the request grows: the split between the two roles is decided on the device)
  // (NB == 2: the second request's chunks follow the first's, so the count is o_indptr[2] = word 7 of the record of the
  // second request's first slot, which is slot o_indptr[1] = word 7 of slot 0's record)
  const int valid_slots = NB == 1 ? a.slot_desc[7] - a.slot_desc[6] : a.slot_desc[8 * a.slot_desc[7] + 7];
  if (slot < valid_slots) {
    fused_decode_attn_workgroup<GROUP, true, NW>(a, slot, kvh);
    return;
  }
  if (ot && threadIdx.x == 0) ot[0] = wall_clock64();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = wave >> 2, w4 = wave & 3;
  const int n_gemv = (g.slots - valid_slots) * a.num_kv_heads;
  const int gi = (slot - valid_slots) + kvh * (g.slots - valid_slots);
  // rows per wave quad; the host's plan keeps enough padding slots that this is <= kOprojMaxRows (the launcher checks the
  // worst case it allows); were it ever larger the surplus rows run as further, un-prefetched passes below
  const int rows_half = (g.M + 2 * n_gemv - 1) / (2 * n_gemv);
  const int row0 = (gi * 2 + quad) * rows_half;
  {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)g.hold_ticks) __builtin_amdgcn_s_sleep(1);
  }
  u32x4 wv[kOprojMaxRows][2];
#pragma unroll
  for (int r = 0; r < kOprojMaxRows; ++r) wv[r][0] = wv[r][1] = u32x4{0u, 0u, 0u, 0u};
  auto request_rows = [&](int base) {
#pragma unroll
    for (int r = 0; r < kOprojMaxRows; ++r) {
      if (base + r >= rows_half) break;
      int row = row0 + base + r;
      row = row < g.M ? row : g.M - 1;   // clamp: loads stay in bounds, stores are masked
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int kk = (w4 + u * 4) * 512 + lane * 8;
        kk = kk < g.K ? kk : g.K - 8;
        wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g.W + (size_t)row * g.K + kk));
      }
    }
  };
  request_rows(0);
  if (ot && threadIdx.x == 0) ot[1] = wall_clock64();
  __shared__ __attribute__((aligned(16))) u32x4 xs[1024];   // K <= 8192
  // per-lane partial dot products, one 64-float line (+4 pad) per (wave quad, row, K-split wave)
  constexpr int kRedStride = 68;
  __shared__ __attribute__((aligned(16))) float red[2 * kOprojMaxRows * 4][kRedStride];
  __shared__ int ok_flag;
  // Per-head-group hand-off (round 5): with GROUP x 128 = 512 a K block of the KSPLIT-4 deal IS one kv head group, so wave
  // w4 needs exactly the attention rows of groups w4 and w4 + 4.  Every wave waits for ITS groups' arrival counters only
  // and takes its own 16 bytes of each straight from global memory (agent-scope loads; the LDS copy of the whole row, its
  // barrier and the wait for the slowest merger before anything starts are gone).  Same arithmetic, same bits.
  const bool group_wait = GROUP == 4 && a.done_stride != 0;   // NB == 2 exists in this form only (launcher)
  u32x4 xg[NB][2];
#pragma unroll
  for (int b = 0; b < NB; ++b) xg[b][0] = xg[b][1] = u32x4{0u, 0u, 0u, 0u};
  if (group_wait) {
    if (threadIdx.x == 0) ok_flag = 1;
    // both arrivals first (lane 0 polls, bounded), then the wave's four 8-byte loads in one round trip
    int ok = 1;
    if (lane == 0) {
      const unsigned long long t0 = wall_clock64();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kb = w4 + u * 4;                       // K block = kv head group
        if (kb * 512 >= g.K) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b)
          while (__hip_atomic_load(a.done_ctr + (size_t)(b * a.num_kv_heads + kb) * a.done_stride, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) < 1) {
            if (wall_clock64() - t0 > g.wait_ticks) { ok = 0; break; }   // 30 ms
            __builtin_amdgcn_s_sleep(1);
          }
      }
      if (!ok && g.status) g.status[0] = 0x300u;       // the host re-runs the step on two launches; this launch's row is void
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    if (ot && threadIdx.x == 0) ot[2] = wall_clock64();
    unsigned long long xl[NB][2][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xl[b][u][0] = xl[b][u][1] = 0ull;
        const int kk = (w4 + u * 4) * 512 + lane * 8;
        if (kk >= g.K) continue;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.o_out + (size_t)b * g.K + kk);
        xl[b][u][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xl[b][u][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        xg[b][u] = ok ? u32x4{(uint32_t)xl[b][u][0], (uint32_t)(xl[b][u][0] >> 32), (uint32_t)xl[b][u][1], (uint32_t)(xl[b][u][1] >> 32)}
                      : u32x4{0u, 0u, 0u, 0u};
    if (ot && threadIdx.x == 0) ot[3] = wall_clock64();
  } else {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      int ok = 1;
      while (__hip_atomic_load(a.done_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.done_target) {
        if (wall_clock64() - t0 > g.wait_ticks) { ok = 0; break; }   // 30 ms
        __builtin_amdgcn_s_sleep(1);
      }
      ok_flag = ok;
      if (!ok && g.status) g.status[0] = 0x300u;
    }
    __syncthreads();
    if (!ok_flag) return;
    if (ot && threadIdx.x == 0) ot[2] = wall_clock64();
    // x = the merged attention row of request 0, written through by the merging workgroups on other CUs / XCDs
    const int nvec = g.K >> 3;
    for (int c = threadIdx.x; c < nvec; c += 512) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.o_out) + (size_t)c * 2;
      const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xs[c] = u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    }
    __syncthreads();
    if (ot && threadIdx.x == 0) ot[3] = wall_clock64();
  }
  for (int base = 0; base < rows_half; base += kOprojMaxRows) {
    if (base > 0) { __syncthreads(); request_rows(base); }   // `red` is reused; only the first pass was prefetched
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b > 0) __syncthreads();                            // `red` is reused by the next request's row
      // all kOprojMaxRows rows are computed, branch-free (rows past rows_half hold zeros / a previous pass: masked below)
      float acc[kOprojMaxRows];
#pragma unroll
      for (int r = 0; r < kOprojMaxRows; ++r) acc[r] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kk = (w4 + u * 4) * 512 + lane * 8;
        const bool live = kk < g.K;
        u32x4 xv = group_wait ? xg[b][u] : xs[live ? (kk >> 3) : 0];
        if (!live) xv = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < kOprojMaxRows; ++r) acc[r] = dot8(wv[r][u], xv, acc[r]);
      }
      if (ot && threadIdx.x == 0 && base == 0 && b == 0) ot[5] = wall_clock64();
      // The 64-lane sums through LDS instead of kOprojMaxRows dependent wave butterflies (1.0 us of the 2.2 us this phase
      // took): every lane parks its partials, then 4 threads per (quad, row, K-split wave) line rebuild wave_sum's own
      // addition tree - lane pairs, quads, 8, 16 inside a thread, the 16-lane rows across the 4 threads - and the four
      // K-split waves are added in order: the bits of gemv_fused_kernel<.., KSPLIT = 4>.
#pragma unroll
      for (int r = 0; r < kOprojMaxRows; ++r) red[(quad * kOprojMaxRows + r) * 4 + w4][lane] = acc[r];
      __syncthreads();
      if (threadIdx.x < 2 * kOprojMaxRows * 16) {
        const int line = threadIdx.x >> 2, t = threadIdx.x & 3;
        const float4* p = reinterpret_cast<const float4*>(&red[line][t * 16]);
        const float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
        const float s0 = (c0.x + c0.y) + (c0.z + c0.w), s1 = (c1.x + c1.y) + (c1.z + c1.w);
        const float s2 = (c2.x + c2.y) + (c2.z + c2.w), s3 = (c3.x + c3.y) + (c3.z + c3.w);
        float v = (s0 + s1) + (s2 + s3);          // one 16-lane row of the wave
        v += dpp_mov<kDppQuadXor1>(v);            // rows 16 apart
        v += dpp_mov<kDppQuadXor2>(v);            // rows 32 apart: the wave's sum, in all 4 threads of the line
        const int rowbase = lane & ~15;           // 16 threads = the 4 K-split lines of one (quad, row)
        const float p0 = __shfl(v, rowbase, kWave), p1 = __shfl(v, rowbase + 4, kWave);
        const float p2 = __shfl(v, rowbase + 8, kWave), p3 = __shfl(v, rowbase + 12, kWave);
        const int qr = threadIdx.x >> 4, q = qr / kOprojMaxRows, r = qr % kOprojMaxRows;
        const int row = (gi * 2 + q) * rows_half + base + r;
        if ((threadIdx.x & 15) == 0 && base + r < rows_half && row < g.M) {
          float y = p0;
          y += p1;
          y += p2;
          y += p3;
          g.Y[(size_t)b * g.M + row] = f2bf(y);
        }
      }
      if (ot && threadIdx.x == 0 && base == 0 && b == 0) ot[6] = wall_clock64();
    }
  }
  if (ot && threadIdx.x == 0) ot[4] = wall_clock64();
}

// merge of the partition-KV partial states: one wave per (request, q head).  Lanes first fetch all
// log2-sum-exps of the request's slots in parallel (<= 64 slots), then every lane accumulates its
// D/64 output dims over the slots with the weights broadcast from registers:
//   out = sum_s 2^(lse_s - M) v_s / sum_s 2^(lse_s - M)
template <int D>
__global__ __launch_bounds__(256) void merge_states_kernel(const Half* __restrict__ tmp_v,
                                                           const float* __restrict__ tmp_s,
                                                           const int* __restrict__ o_indptr,
                                                           Half* __restrict__ out, int batch_size,
                                                           int num_qo_heads) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= batch_size * num_qo_heads) return;
  const int b = unit / num_qo_heads, head = unit - b * num_qo_heads;
  merge_one<D, false>(tmp_v, tmp_s, o_indptr[b], o_indptr[b + 1], head, num_qo_heads,
               out + ((size_t)b * num_qo_heads + head) * D);
}

static unsigned long long* g_attn_trace = nullptr;  // pegainfer_debug_attn_trace

static void fill_args(DecodeAttnArgs& a, const Half* q, Half* output, const Half* kv, long k_off, long v_off,
                      const int* pi, const int* pip, const int* lpl, const int* ri, const int* kti, const int* kcs,
                      const uint8_t* mask, Half* tmp_v, float* tmp_s, int hq, int hkv, int page_size,
                      long stride_page, float sm_scale) {
  a = DecodeAttnArgs{};
  a.q = q; a.o_out = output; a.kv = kv; a.k_off = k_off; a.v_off = v_off;
  a.page_indices = pi; a.page_indptr = pip; a.last_page_len = lpl; a.request_indices = ri;
  a.kv_tile_indices = kti; a.kv_chunk_size_ptr = kcs; a.block_valid_mask = mask; a.tmp_v = tmp_v; a.tmp_s = tmp_s;
  a.num_qo_heads = hq; a.num_kv_heads = hkv; a.page_size = page_size; a.stride_page = stride_page;
  a.scale_log2 = sm_scale * 1.4426950408889634f;
  a.trace = g_attn_trace;
}

template <int D, bool PARTITION, bool FUSED>
static int launch_decode(const DecodeAttnArgs& a, const int* o_indptr, int batch_size, int slots, hipStream_t s) {
  if (slots <= 0 || a.num_kv_heads <= 0) return 0;
  if (o_indptr && PARTITION && slots > batch_size * 64) return static_cast<int>(hipErrorInvalidValue);
  const int group = a.num_qo_heads / a.num_kv_heads;
  dim3 grid(slots, a.num_kv_heads);
  // 8 waves per workgroup when the launch has at most one workgroup per CU (bs 32 un-split: 4.60 -> 4.42 ms/step,
  // bs 1 at ctx 10000: 2.82 -> 2.67): twice the loads in flight per CU.  With two workgroups per CU the 64 KB of
  // partial states per workgroup cost more than they give (bs 8 / 16: +9 %).  PEGAINFER_ATTN_WAVES = 4 | 8 forces one.
  static const int nw_env = [] { const char* e = getenv("PEGAINFER_ATTN_WAVES"); return e && *e ? atoi(e) : 0; }();
  const bool wide = nw_env == 8 || (nw_env == 0 && (long)slots * a.num_kv_heads <= 256);
#define PK_LAUNCH(G)                                                                                \
  do {                                                                                              \
    if (wide) {                                                                                     \
      if constexpr (FUSED) fused_decode_attn_kernel<G, PARTITION, 8><<<grid, 512, 0, s>>>(a);       \
      else decode_attn_kernel<D, G, PARTITION, 8><<<grid, 512, 0, s>>>(a);                          \
    } else {                                                                                        \
      if constexpr (FUSED) fused_decode_attn_kernel<G, PARTITION, 4><<<grid, 256, 0, s>>>(a);       \
      else decode_attn_kernel<D, G, PARTITION, 4><<<grid, 256, 0, s>>>(a);                          \
    }                                                                                               \
  } while (0)
  switch (group) {
    case 1: PK_LAUNCH(1); break;
    case 2: PK_LAUNCH(2); break;
    case 4: PK_LAUNCH(4); break;
    case 8: PK_LAUNCH(8); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
#undef PK_LAUNCH
  if (PARTITION && !a.merge_counters)
    merge_states_kernel<D><<<ceil_div((long)batch_size * a.num_qo_heads, 4), 256, 0, s>>>(
        a.tmp_v, a.tmp_s, o_indptr, a.o_out, batch_size, a.num_qo_heads);
  return static_cast<int>(hipGetLastError());
}

}  // namespace pk

using namespace pk;

extern "C" {

// Debug: device buffer of slots * kv_heads * 8 uint64 that every later fused decode-attention launch stamps with the
// 100 MHz wall clock at its phase boundaries (entry, record read, q prologue, scan, partials published, ticket,
// merge; word 7 = 1 for the merging workgroup); nullptr switches it off.  Not part of the reference ABI.
void pegainfer_debug_attn_trace(uint64_t* buf) { g_attn_trace = reinterpret_cast<unsigned long long*>(buf); }

int32_t paged_attention_decode_cuda(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems,
                                    int64_t v_offset_elems, const int32_t* page_indices,
                                    const int32_t* page_indptr, const int32_t* last_page_len_d,
                                    const int32_t* request_indices, const int32_t* kv_tile_indices,
                                    const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads, int32_t num_kv_heads,
                                    int32_t head_dim, int32_t page_size, int32_t batch_size, int64_t stride_page,
                                    float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128) return static_cast<int32_t>(hipErrorInvalidValue);  // HEAD_DIM=128 instantiation
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, nullptr, nullptr, nullptr, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<128, false, false>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

int32_t paged_attention_decode_split_kv_cuda(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr,
    const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size,
    int64_t stride_page, float sm_scale, pegainfer_stream_t stream) {
  if (head_dim != 128) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, block_valid_mask, tmp_v, tmp_s, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<128, true, false>(a, o_indptr, batch_size, padded_batch_size, as_stream(stream));
}

// HEAD_DIM = 256 instantiation (Qwen3.5 full-attention layers, ffi.rs:1286-1306): 32 lanes per K row,
// 2 rows per load instruction, otherwise the same kernel.
int32_t paged_attention_decode_cuda_hd256(const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems,
                                          int64_t v_offset_elems, const int32_t* page_indices,
                                          const int32_t* page_indptr, const int32_t* last_page_len_d,
                                          const int32_t* request_indices, const int32_t* kv_tile_indices,
                                          const int32_t* kv_chunk_size_ptr, int32_t num_qo_heads,
                                          int32_t num_kv_heads, int32_t head_dim, int32_t page_size,
                                          int32_t batch_size, int64_t stride_page, float sm_scale,
                                          pegainfer_stream_t stream) {
  if (head_dim != 256) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, nullptr, nullptr, nullptr, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  return launch_decode<256, false, false>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

// Extension: partition-KV decode at head_dim 256.  The reference has no split symbol for the Qwen3.5 full-attention
// layers (ffi.rs:1286-1306 is non-partition only), which leaves bs = 1 with num_kv_heads = 4 workgroups on a
// 256-CU part; same plan arrays and scratch contract as paged_attention_decode_split_kv_cuda.
int32_t pegainfer_paged_attention_decode_split_kv_hd256(
    const Half* q, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* request_indices, const int32_t* kv_tile_indices, const int32_t* kv_chunk_size_ptr,
    const int32_t* o_indptr, const uint8_t* block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t padded_batch_size,
    int64_t stride_page, float sm_scale, int32_t* merge_counters, pegainfer_stream_t stream) {
  if (head_dim != 256) return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, q, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            request_indices, kv_tile_indices, kv_chunk_size_ptr, block_valid_mask, tmp_v, tmp_s, num_qo_heads,
            num_kv_heads, page_size, stride_page, sm_scale);
  a.merge_counters = merge_counters;  // optional in-launch merge, as in pegainfer_fused_decode_attention
  a.o_indptr = o_indptr;
  return launch_decode<256, true, false>(a, o_indptr, batch_size, padded_batch_size, as_stream(stream));
}

// Extension (include/pegainfer_kernels_ext.h): qk_norm_rope_batched_decode_cuda + paged_kv_scatter_cuda +
// paged_attention_decode[_split_kv]_cuda in one launch (+ the merge when partitioned).
int32_t pegainfer_fused_decode_attention(
    const Half* qkv, Half* output, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* positions, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache,
    const Half* sin_cache, float rms_eps, int32_t use_split, const int32_t* split_request_indices,
    const int32_t* split_kv_tile_indices, const int32_t* split_kv_chunk_size_ptr, const int32_t* split_o_indptr,
    const uint8_t* split_block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t split_slots, int64_t stride_page,
    float sm_scale, const int32_t* slot_desc, int32_t* merge_counters, pegainfer_stream_t stream) {
  if (head_dim != 128 || !host_aligned16(qkv) || !host_aligned16(kv_data) || !host_aligned16(slot_desc))
    return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, nullptr, output, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            use_split ? split_request_indices : nullptr, split_kv_tile_indices, split_kv_chunk_size_ptr,
            use_split ? split_block_valid_mask : nullptr, tmp_v, tmp_s, num_qo_heads, num_kv_heads, page_size,
            stride_page, sm_scale);
  a.qkv = qkv; a.q_norm_w = q_norm_weight; a.k_norm_w = k_norm_weight; a.cos_cache = cos_cache;
  a.sin_cache = sin_cache; a.positions = positions; a.eps = rms_eps; a.slot_desc = slot_desc;
  if (use_split) {
    a.merge_counters = merge_counters; a.o_indptr = split_o_indptr;
    return launch_decode<128, true, true>(a, split_o_indptr, batch_size, split_slots, as_stream(stream));
  }
  return launch_decode<128, false, true>(a, nullptr, batch_size, batch_size, as_stream(stream));
}

// Extension: pegainfer_fused_decode_attention (partition form) + the o_proj GEMV behind it in ONE launch, for steps with
// one request (see attn_oproj_kernel).  `o_proj` [hidden, q_dim] row-major, `attn_proj_out` [hidden]; `attn_out` still
// receives the attention row (the o_proj phase reads it from there).  done_counter: one device int, ZERO before the
// launch (the host runtime keeps one per layer and clears them once per step).  Returns hipErrorInvalidValue when the
// shape does not fit the form (then the caller runs the two stand-alone launches): batch_size != 1, more attention
// workgroups than CUs, q_dim not a multiple of 2048 or > 8192, more than 10 o_proj rows per CU.
// Shape test of the fused attention + o_proj launch, by configuration only (no pointers): a host runtime asks ONCE, at model
// creation, whether its steps of `batch_size` requests can take the form, and plans its KV chunks accordingly from the first
// step on (before round 5 the launcher's refusal inside the first graph capture was the only way to find out).
static bool oproj_shape_fits(int num_qo_heads, int num_kv_heads, int head_dim, int hidden, int split_slots, int min_padding_slots,
                             int batch_size) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  const int q_dim = num_qo_heads * head_dim;
  const int grid = split_slots * num_kv_heads;
  const int n_gemv = min_padding_slots * num_kv_heads;
  const int group = num_kv_heads > 0 ? num_qo_heads / num_kv_heads : 0;
  if (head_dim != 128 || batch_size < 1 || batch_size > 2 || split_slots < 1 || min_padding_slots < 1 ||
      min_padding_slots >= split_slots || grid > cus || (q_dim % 2048) != 0 || q_dim / 2048 > 2 /* 2 K blocks per wave */ ||
      hidden < 1 || (group != 1 && group != 2 && group != 4))
    return false;
  if (n_gemv < 1 || ceil_div(hidden, 2 * n_gemv) > kOprojMaxRows) return false;
  if (batch_size == 2 && !(group == 4 && q_dim / 512 == num_kv_heads)) return false;   // two requests: per-group form only
  return true;
}
int32_t pegainfer_fused_decode_attention_oproj_supported(int32_t num_qo_heads, int32_t num_kv_heads, int32_t head_dim, int32_t hidden,
                                                         int32_t split_slots, int32_t min_padding_slots, int32_t batch_size) {
  return oproj_shape_fits(num_qo_heads, num_kv_heads, head_dim, hidden, split_slots, min_padding_slots, batch_size) ? 1 : 0;
}

int32_t pegainfer_fused_decode_attention_oproj(
    const Half* qkv, Half* attn_out, const Half* kv_data, int64_t k_offset_elems, int64_t v_offset_elems,
    const int32_t* page_indices, const int32_t* page_indptr, const int32_t* last_page_len_d,
    const int32_t* positions, const Half* q_norm_weight, const Half* k_norm_weight, const Half* cos_cache,
    const Half* sin_cache, float rms_eps, const int32_t* split_request_indices,
    const int32_t* split_kv_tile_indices, const int32_t* split_kv_chunk_size_ptr, const int32_t* split_o_indptr,
    const uint8_t* split_block_valid_mask, Half* tmp_v, float* tmp_s, int32_t num_qo_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t page_size, int32_t batch_size, int32_t split_slots, int32_t min_padding_slots,
    int64_t stride_page, float sm_scale, const int32_t* slot_desc, int32_t* merge_counters, const Half* o_proj,
    Half* attn_proj_out, int32_t hidden, int32_t* done_counter, uint32_t* status, pegainfer_stream_t stream) {
  const int q_dim = num_qo_heads * head_dim;
  const int grid = split_slots * num_kv_heads;
  const int group = num_kv_heads > 0 ? num_qo_heads / num_kv_heads : 0;
  if (!oproj_shape_fits(num_qo_heads, num_kv_heads, head_dim, hidden, split_slots, min_padding_slots, batch_size) ||
      !merge_counters || !done_counter || !o_proj || !attn_proj_out || !slot_desc || !host_aligned16(qkv) ||
      !host_aligned16(kv_data) || !host_aligned16(slot_desc) || !host_aligned16(o_proj) || !host_aligned16(attn_out))
    return static_cast<int32_t>(hipErrorInvalidValue);
  DecodeAttnArgs a;
  fill_args(a, nullptr, attn_out, kv_data, k_offset_elems, v_offset_elems, page_indices, page_indptr, last_page_len_d,
            split_request_indices, split_kv_tile_indices, split_kv_chunk_size_ptr, split_block_valid_mask, tmp_v, tmp_s,
            num_qo_heads, num_kv_heads, page_size, stride_page, sm_scale);
  a.qkv = qkv; a.q_norm_w = q_norm_weight; a.k_norm_w = k_norm_weight; a.cos_cache = cos_cache;
  a.sin_cache = sin_cache; a.positions = positions; a.eps = rms_eps; a.slot_desc = slot_desc;
  a.merge_counters = merge_counters; a.o_indptr = split_o_indptr; a.done_ctr = done_counter;
  // per-head-group arrival counters (done_counter = num_kv_heads ints, kMergeCtrStride apart) when a K block of the
  // o_proj deal is one kv head group; PEGAINFER_OPROJ_GROUPWAIT=0 keeps the single counter + whole-row wait (A/B, same bits)
  static const bool gw = [] { const char* e = getenv("PEGAINFER_OPROJ_GROUPWAIT"); return !(e && e[0] == '0'); }();
  a.done_stride = (gw || batch_size == 2) && group == 4 && q_dim / 512 == num_kv_heads ? kMergeCtrStride : 0;
  static const int hold = [] { const char* e = getenv("PEGAINFER_ATTN_OPROJ_HOLD"); return e && *e ? atoi(e) : 0; }();
  static const unsigned long long wait_ticks = [] { const char* e = getenv("PEGAINFER_OPROJ_WAIT_TICKS"); return e && *e ? strtoull(e, nullptr, 10) : 3000000ull; }();
  OprojArgs g{o_proj, attn_proj_out, hidden, q_dim, split_slots, num_kv_heads * batch_size, status, hold, wait_ticks};
  hipStream_t s = as_stream(stream);
  if (batch_size == 2) {
    attn_oproj_kernel<4, 2><<<grid, 512, 0, s>>>(a, g);
  } else {
    switch (group) {
      case 1: attn_oproj_kernel<1, 1><<<grid, 512, 0, s>>>(a, g); break;
      case 2: attn_oproj_kernel<2, 1><<<grid, 512, 0, s>>>(a, g); break;
      default: attn_oproj_kernel<4, 1><<<grid, 512, 0, s>>>(a, g); break;
    }
  }
  return static_cast<int32_t>(hipGetLastError());
}

}  // extern "C"
