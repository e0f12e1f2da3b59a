// Persistent decode-step engine for gfx950 (bs = 1): all layers of one Qwen3 decode step in ONE launch.
//
// Why: at ~2 ms per step the fused path (decode_mode 1) is five weight-streaming kernels per layer whose fixed cost
// (launch boundary + first-byte latency + tail, ~4-7 us each) is half of the layer time; no kernel can hide it because
// nothing of the NEXT matrix can be in flight while the previous kernel drains.  Here every CU runs one persistent
// workgroup = 1 loader wave + 3 consumer waves.  The loader streams this workgroup's row slice of EVERY matrix of the
// step, in op order, into a ring of 16 KiB LDS slots with LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction,
// non-temporal, three fills in flight); it depends on no activation, only on ring space, so the weights of the next
// op keep arriving while the consumers wait for the previous op's all-to-all hand-off.  The consumers take pieces out
// of the ring (ds_read_b128 + v_dot2c against the LDS-resident x) and hand each op's output vector to every other
// workgroup through write-through (sc1) stores + sharded arrival counters; readers use sc1 loads, nobody fences.
//
// Ops per layer: QKV GEMV (previous residual add + RMSNorm in the x staging) -> paged decode attention with q/k norm +
// RoPE + KV append (partition-KV partials merged by the last workgroup of each kv head) -> O GEMV -> gate|up GEMV
// (add + RMSNorm staging, SwiGLU epilogue) -> down GEMV.  Every arithmetic core is shared with the kernels of
// decode_mode 1 (norm_core.h, rope_core.h, attn_decode_core.h, the per-lane dot2 chains and wave butterflies of
// gemv_core.h restated for one consumer wave emulating the four K-split waves), so logits are bit-identical to
// decode_mode 0 / 1 (tests/test_gpu_real_dims.py::test_engine_*).
//
// Correctness does not depend on placement or dispatch order: one workgroup per CU by LDS size, grid <= CU count, every
// spin bounded (status word != 0 -> all waves leave, the host falls back to decode_mode 1).
#include "attn_decode_core.h"
#include "norm_core.h"
#include "pegainfer_kernels_ext.h"

namespace pk {

typedef pegainfer_engine_args_t EngArgs;

constexpr int kEngSlots = 5;                               // ring slots
constexpr int kFillPieces = 16;                            // 1 KiB pieces per fill (= one slot)
constexpr int kSlotBytes = kFillPieces * 1024;
constexpr int kRingBytes = kEngSlots * kSlotBytes;         // 80 KiB
constexpr int kAttnParts = 32;                             // 8 virtual waves x 4 lane rows
constexpr int kScratchBytes = kAttnParts * 4 * (128 + 2) * 4;  // attention partial states (GROUP 4); x vector aliases it
constexpr int kOutsBytes = 512;                            // up to 256 bf16 outputs of this workgroup per op
constexpr int kSyncBytes = 64;
constexpr int kEngMaxLayers = 64;
constexpr int kTableBytes = kEngMaxLayers * 64;            // the per-layer weight pointers, copied to LDS at kernel start
constexpr int kEngLdsBytes = kRingBytes + kScratchBytes + kOutsBytes + kSyncBytes + kTableBytes;
constexpr int kFillsInFlight = 3;                          // 48 loads <= the 6-bit vmcnt

enum { kOpQkv = 0, kOpAttn = 1, kOpO = 2, kOpGu = 3, kOpDown = 4, kEngOps = 5, kEngShards = 8 };
// LDS sync words
enum { kSyReady = 0, kSyDone0 = 1, kSyBar = 4, kSyFlag = 5 };
// status codes (status[0]); status[1] = where
enum { kEngErrLoaderFree = 1, kEngErrReady = 2, kEngErrBarrier = 3, kEngErrPoll = 4 };

constexpr unsigned kSpinLimit = 1u << 22;

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_v4;

struct EngCtx {
  const EngArgs* a;
  lds_u8* lds;                  // dynamic LDS base (explicit address space: every flag / ring access must be a DS op -
                                // a flat access in the loader would count on vmcnt next to the hand-counted DMAs)
  lds_u32* sy;                  // sync words
  const lds_u8* ltab;           // layer table copy
  int wg, nwg, lane, cw;        // cw: consumer wave 0..2 (loader: -1)
  unsigned bar_gen;             // consumer barrier generation (per wave)
  unsigned ready_seen;          // cached count of landed fills
  bool dead;
  unsigned long long t_wait;    // trace: cycles spent waiting (loader: for a free slot; consumer: for a landed fill)
};
__device__ __forceinline__ unsigned long long eng_now() { return __builtin_amdgcn_s_memtime(); }

__device__ __forceinline__ bool eng_aborted(const EngArgs& a) {
  return __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __noinline__ void eng_fail_record(uint32_t* status, unsigned code, unsigned where) {
  // first failure wins: atomic max on a zero word keeps the code, the location goes next to it
  if (__hip_atomic_fetch_max(status, code | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
    __hip_atomic_store(status + 1, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void eng_fail(EngCtx& c, unsigned code, unsigned where) {
  if (c.lane == 0) eng_fail_record(c.a->status, code, where);
  c.dead = true;
}
// one slow-path check per 1024 spins: give up after kSpinLimit or when any wave of the grid gave up
__device__ __forceinline__ bool eng_spin_check(EngCtx& c, unsigned& spins, unsigned code, unsigned where) {
  __builtin_amdgcn_s_sleep(2);
  if ((++spins & 1023u) == 0u) {
    if (spins >= kSpinLimit) { eng_fail(c, code, where); return false; }
    if (eng_aborted(*c.a)) { c.dead = true; return false; }
  }
  return true;
}
// LDS flags: relaxed atomics + compiler barriers.  The LDS pipeline executes one wave's DS operations in order, which
// is all the ordering the flags need; release / acquire orderings would make hipcc add s_waitcnt vmcnt(0), draining the
// loader's DMAs.
__device__ __forceinline__ unsigned lds_load_acq(lds_u32* p) {
  const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void lds_store_rel(lds_u32* p, unsigned v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- barrier among the three consumer waves (the loader never takes part) ----
__device__ __forceinline__ bool eng_cbar(EngCtx& c) {
  c.bar_gen += 1;
  asm volatile("" ::: "memory");
  if (c.lane == 0) __hip_atomic_fetch_add(c.sy + kSyBar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  while (lds_load_acq(c.sy + kSyBar) < c.bar_gen * 3u)
    if (!eng_spin_check(c, spins, kEngErrBarrier, c.bar_gen)) return false;
  return true;
}

// ---- arrival / wait on an op's sharded counters (global, agent scope) ----
// The 8 shard counters of an (layer, op) edge sit kEngCtrStride words apart - a cache line each: arrivals on eight
// counters that share one line serialise like arrivals on one (measured on the GEMV ticket counters: 1024 atomics on
// one line = 12 us).
constexpr int kEngCtrStride = 32;
__device__ __forceinline__ unsigned* eng_ctr(const EngArgs& a, int layer, int op) {
  return a.sync + ((size_t)layer * kEngOps + op) * kEngShards * kEngCtrStride;
}
__device__ __forceinline__ void eng_arrive(EngCtx& c, int layer, int op, int shard) {
  // caller: every wave that stored payload has drained it (s_waitcnt vmcnt(0)) and met at a consumer barrier
  if (c.lane == 0)
    __hip_atomic_fetch_add(eng_ctr(*c.a, layer, op) + shard * kEngCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer wave 0 polls the 8 words (one 32-byte sc1 load per poll); everybody then meets at the barrier
__device__ __forceinline__ bool eng_wait_op(EngCtx& c, int layer, int op, unsigned per_shard, unsigned shard0_only) {
  if (c.cw == 0) {
    const unsigned* p = eng_ctr(*c.a, layer, op);
    const unsigned want = shard0_only ? (c.lane == 0 ? shard0_only : 0u) : per_shard;
    unsigned spins = 0;
    for (;;) {
      unsigned v = want;
      if (c.lane < kEngShards && !(shard0_only && c.lane != 0))
        v = __hip_atomic_load(p + c.lane * kEngCtrStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(v >= want)) break;
      if (!eng_spin_check(c, spins, kEngErrPoll, (unsigned)(layer * kEngOps + op))) break;
    }
  }
  return eng_cbar(c) && !c.dead;
}

// 16-byte write-through (sc1) loads of activations handed over inside the launch
__device__ __forceinline__ u32x4 ld_act16(const __amdgpu_buffer_rsrc_t& rs, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
}
__device__ __forceinline__ u32x4 ld_act16_plain(const __amdgpu_buffer_rsrc_t& rs, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// =====================================================================================================================
// loader wave
// =====================================================================================================================
struct EngShape {   // stream ops in order: 0 qkv, 1 o, 2 gate|up (two rows per pair), 3 down
  int H, I, QD, KVD;
  int rows0, rows1, rows2, rows3, ppr0, ppr1, ppr2, ppr3, np0, np1, np2, np3, per_layer;
  // selects, not arrays: a runtime-indexed array would live in scratch memory, i.e. behind counted VMEM loads
  __device__ __forceinline__ int rows(int op) const { return op == 0 ? rows0 : op == 1 ? rows1 : op == 2 ? rows2 : rows3; }
  __device__ __forceinline__ int ppr(int op) const { return op == 0 ? ppr0 : op == 1 ? ppr1 : op == 2 ? ppr2 : ppr3; }
};
__device__ __forceinline__ EngShape eng_shape(const EngArgs& a, int nwg) {
  EngShape s;
  s.H = a.hidden; s.I = a.intermediate; s.QD = a.num_qo_heads * a.head_dim; s.KVD = a.num_kv_heads * a.head_dim;
  s.rows0 = (s.QD + 2 * s.KVD) / nwg; s.ppr0 = s.H / 512;
  s.rows1 = s.H / nwg;                s.ppr1 = s.QD / 512;
  s.rows2 = 2 * (s.I / nwg);          s.ppr2 = s.H / 512;
  s.rows3 = s.H / nwg;                s.ppr3 = s.I / 512;
  s.np0 = s.rows0 * s.ppr0; s.np1 = s.rows1 * s.ppr1; s.np2 = s.rows2 * s.ppr2; s.np3 = s.rows3 * s.ppr3;
  s.per_layer = s.np0 + s.np1 + s.np2 + s.np3;
  return s;
}
// layer table entry in LDS: 8 pointers in the order of pegainfer_engine_layer_t {qkv, o, gate_up, down, ln1, ln2, q_norm, k_norm}
__device__ __forceinline__ const unsigned char* eng_tab_ptr(const EngCtx& c, int layer, int field) {
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  const unsigned long long v = *reinterpret_cast<const lds_u64*>(c.ltab + layer * 64 + field * 8);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  // through the global address space so that loads from it are global_load, never flat_load
  typedef const __attribute__((address_space(1))) unsigned char* gptr_t;
  return (const unsigned char*)(gptr_t)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const unsigned char* eng_row_ptr(const EngCtx& c, const EngShape& s, int layer, int sop, int sr, int wg) {
  switch (sop) {
    case 0: return eng_tab_ptr(c, layer, 0) + ((size_t)wg * s.rows0 + sr) * s.H * 2;
    case 1: return eng_tab_ptr(c, layer, 1) + ((size_t)wg * s.rows1 + sr) * s.QD * 2;
    case 2: {  // pair p = sr >> 1: gate row, then up row
      const size_t row = (size_t)wg * (s.rows2 >> 1) + (sr >> 1) + ((sr & 1) ? (size_t)s.I : 0);
      return eng_tab_ptr(c, layer, 2) + row * s.H * 2;
    }
    default: return eng_tab_ptr(c, layer, 3) + ((size_t)wg * s.rows3 + sr) * s.I * 2;
  }
}

__device__ __forceinline__ void eng_loader(EngCtx& c) {
  const EngArgs& a = *c.a;
  const EngShape s = eng_shape(a, c.nwg);
  const long total_pieces = (long)a.layers * s.per_layer;
  const int total_fills = (int)((total_pieces + kFillPieces - 1) / kFillPieces);
  const unsigned ring_lds = (unsigned)(uintptr_t)c.lds;
  int layer = 0, sop = 0, sr = 0, kb = 0, cur_ppr = s.ppr0, cur_rows = s.rows0;
  const unsigned long long t_begin = eng_now();
  long issued = 0;
  const unsigned char* row = eng_row_ptr(c, s, 0, 0, 0, c.wg);
  const unsigned lane_off = (unsigned)c.lane * 16u;
  for (int fill = 0; fill < total_fills; ++fill) {
    if (fill >= kEngSlots) {  // the slot still holds fill - kEngSlots: every consumer must be past its pieces
      const unsigned need = (unsigned)(fill - kEngSlots + 1) * kFillPieces;
      unsigned spins = 0;
      const unsigned long long tw0 = a.trace ? eng_now() : 0ull;
      for (;;) {
        const unsigned d0 = lds_load_acq(c.sy + kSyDone0), d1 = lds_load_acq(c.sy + kSyDone0 + 1),
                       d2 = lds_load_acq(c.sy + kSyDone0 + 2);
        const unsigned m = d0 < d1 ? (d0 < d2 ? d0 : d2) : (d1 < d2 ? d1 : d2);
        if (m >= need) break;
        if (!eng_spin_check(c, spins, kEngErrLoaderFree, (unsigned)fill)) return;
      }
      if (a.trace && spins) c.t_wait += eng_now() - tw0;
    }
    const unsigned slot_lds = ring_lds + (unsigned)(fill % kEngSlots) * kSlotBytes;
#pragma unroll
    for (int j = 0; j < kFillPieces; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(slot_lds + (unsigned)j * 1024u);
      const unsigned voff = lane_off + (unsigned)kb * 1024u;
      const unsigned long long base = (unsigned long long)row;
      const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
      const unsigned long long sbase = ((unsigned long long)bhi << 32) | blo;
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                   :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
      if (issued + 1 < total_pieces) {  // after the last real piece the remaining loads of the fill repeat it
        ++issued;
        if (++kb == cur_ppr) {
          kb = 0;
          if (++sr == cur_rows) {
            sr = 0;
            if (++sop == 4) { sop = 0; ++layer; }
            cur_ppr = s.ppr(sop);
            cur_rows = s.rows(sop);
          }
          row = eng_row_ptr(c, s, layer, sop, sr, c.wg);
        }
      }
    }
    // the oldest of the kFillsInFlight fills has landed once at most (kFillsInFlight - 1) * 16 loads are outstanding
    if (fill >= kFillsInFlight - 1) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kFillsInFlight - 1) * kFillPieces) : "memory");
      lds_store_rel(c.sy + kSyReady, (unsigned)(fill - (kFillsInFlight - 1) + 1));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_store_rel(c.sy + kSyReady, (unsigned)total_fills);
  if (a.trace && c.lane == 0) {
    a.trace[(size_t)c.wg * 32 + 20] = c.t_wait;
    a.trace[(size_t)c.wg * 32 + 21] = eng_now() - t_begin;
  }
}

// =====================================================================================================================
// consumer waves
// =====================================================================================================================
__device__ __forceinline__ void eng_publish_done(EngCtx& c, unsigned piece) {
  if (c.lane == 0) lds_store_rel(c.sy + kSyDone0 + c.cw, piece);
}
__device__ __forceinline__ bool eng_need_fill(EngCtx& c, unsigned fill) {
  if (fill < c.ready_seen) return true;
  unsigned spins = 0;
  const unsigned long long tw0 = c.a->trace ? eng_now() : 0ull;
  for (;;) {
    c.ready_seen = lds_load_acq(c.sy + kSyReady);
    if (fill < c.ready_seen) break;
    if (!eng_spin_check(c, spins, kEngErrReady, fill)) return false;
  }
  if (c.a->trace) c.t_wait += eng_now() - tw0;
  return true;
}
__device__ __forceinline__ u32x4 eng_piece(const EngCtx& c, unsigned g) {
  const unsigned fill = g >> 4;
  const unsigned off = (fill % kEngSlots) * kSlotBytes + (g & 15u) * 1024u + (unsigned)c.lane * 16u;
  return *reinterpret_cast<const lds_v4*>(c.lds + off);
}

// dot product of one weight row (pieces g0 .. g0 + ppr - 1 of the stream) with the LDS-resident x.  KS4 restates the
// K split of gemv_fused_kernel<..., KSPLIT = 4>: K block b goes to virtual wave b & 3, each virtual wave is one dot2
// chain per lane in block order, wave butterfly per virtual wave, partials summed in wave order.  !KS4 = KSPLIT 1.
template <bool KS4>
__device__ __forceinline__ bool eng_row_dot(EngCtx& c, const u32x4* xs, unsigned g0, int ppr, float& out) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < ppr; kb += 4) {
    if (!eng_need_fill(c, (g0 + kb + (ppr - kb < 4 ? ppr - kb : 4) - 1) >> 4)) return false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kb + j < ppr) {
        const u32x4 w = eng_piece(c, g0 + kb + j);
        const u32x4 x = xs[(kb + j) * 64 + c.lane];
        if (KS4) acc[j] = dot8(w, x, acc[j]);
        else acc[0] = dot8(w, x, acc[0]);
      }
    }
  }
  if (KS4) {
    float v = wave_sum(acc[0]);
    v += wave_sum(acc[1]);
    v += wave_sum(acc[2]);
    v += wave_sum(acc[3]);
    out = v;
  } else {
    out = wave_sum(acc[0]);
  }
  return true;
}

// x = rms_norm(X [+ R]) * w into LDS (canonical order of norm_core.h: every consumer wave reduces the whole row
// itself, lane l folding vectors l, l + 64, ...), hidden_out = bf16(X + R) by workgroup 0.
__device__ __forceinline__ void eng_stage_norm(EngCtx& c, u32x4* xs, const Half* X, bool x_coherent, const Half* R,
                                               const Half* w, Half* hidden_out, int H, float eps) {
  const int nvec = H >> 3;
  const __amdgpu_buffer_rsrc_t xr = act_rsrc(X, H * 2), rr = act_rsrc(R ? R : X, H * 2);
  constexpr int MAXV = 8;   // H <= 4096
  u32x4 hv[MAXV], rv[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = c.lane + 64 * j;
    if (i < nvec) {
      hv[j] = x_coherent ? ld_act16(xr, i * 16) : ld_act16_plain(xr, i * 16);
      if (R) rv[j] = ld_act16(rr, i * 16);
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = c.lane + 64 * j;
    if (i < nvec) {
      if (R) add_sq8(hv[j], rv[j], ss);
      else sq8(hv[j], ss);
    }
  }
  ss = wave_sum(ss);
  const float inv = rsqrtf(__fadd_rn(ss / (float)H, eps));
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = c.lane + 64 * j;
    if (i < nvec && (j % 3) == c.cw) {
      const u32x4 gw = reinterpret_cast<const u32x4*>(w)[i];
      u32x4 nh;
      xs[i] = R ? norm_scale8(hv[j], &rv[j], gw, inv, 0.f, &nh) : norm_scale8(hv[j], nullptr, gw, inv, 0.f, nullptr);
      if (R && hidden_out && c.wg == 0) {
        Half* dst = hidden_out + (size_t)i * 8;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(nh) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void eng_stage_plain(EngCtx& c, u32x4* xs, const Half* X, int K) {
  const int nvec = K >> 3;
  const __amdgpu_buffer_rsrc_t xr = act_rsrc(X, K * 2);
  for (int i0 = c.cw * 64; i0 < nvec; i0 += 4 * 192) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 192 + c.lane;
      if (i < nvec) v[u] = ld_act16(xr, i * 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 192 + c.lane;
      if (i < nvec) xs[i] = v[u];
    }
  }
}

// rows (or gate/up pairs) cw, cw + 3, ... of this workgroup's slice -> bf16 bits in the LDS out array
template <bool KS4, bool SILU>
__device__ __forceinline__ bool eng_gemv_rows(EngCtx& c, const u32x4* xs, uint16_t* outs, unsigned gop, int units, int ppr,
                                              unsigned gnext) {
  const int upp = SILU ? 2 * ppr : ppr;   // pieces per unit
  for (int r = c.cw; r < units; r += 3) {
    const unsigned g0 = gop + (unsigned)r * upp;
    eng_publish_done(c, g0);
    float v0, v1 = 0.f;
    if (!eng_row_dot<KS4>(c, xs, g0, ppr, v0)) return false;
    if (SILU && !eng_row_dot<KS4>(c, xs, g0 + ppr, ppr, v1)) return false;
    if (c.lane == 0) {
      if (SILU) {
        const float gt = bf16_round_f(v0), up = bf16_round_f(v1);  // the GEMM output is bf16 before SwiGLU
        outs[r] = f2bf(silu_f(gt) * up);
      } else {
        outs[r] = f2bf(v0);
      }
    }
  }
  eng_publish_done(c, gnext);
  return true;
}
// out array -> global (write-through), then this workgroup's arrival
__device__ __forceinline__ bool eng_store_out(EngCtx& c, const uint16_t* outs, Half* Y, int units, int layer, int op) {
  if (!eng_cbar(c)) return false;
  if (c.cw == 0) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(Y + (size_t)c.wg * units);
    for (int i = c.lane; i < units / 2; i += 64)
      __hip_atomic_store(dst + i, (uint32_t)outs[2 * i] | ((uint32_t)outs[2 * i + 1] << 16), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    eng_arrive(c, layer, op, c.wg & (kEngShards - 1));
  }
  return true;
}

// ---- attention phase of one layer: (slot, kv head) items over the workgroups ----
template <int GROUP>
__device__ __forceinline__ bool eng_attention(EngCtx& c, int layer, float* scratch) {
  const EngArgs& a = *c.a;
  constexpr int D = 128;
  typedef AttnScan<D, GROUP> Scan;
  float* sm_m = scratch;
  float* sm_l = scratch + kAttnParts * GROUP;
  float* sm_o = scratch + 2 * kAttnParts * GROUP;
  const Half* q_norm_w = reinterpret_cast<const Half*>(eng_tab_ptr(c, layer, 6));
  const Half* k_norm_w = reinterpret_cast<const Half*>(eng_tab_ptr(c, layer, 7));
  DecodeAttnArgs da = DecodeAttnArgs{};
  da.o_out = a.attn_out; da.kv = a.kv_data; da.k_off = (long)layer * a.layer_stride; da.v_off = da.k_off + a.kv_block_len;
  da.page_indices = a.page_indices; da.tmp_v = a.tmp_v; da.tmp_s = a.tmp_s; da.num_qo_heads = a.num_qo_heads;
  da.num_kv_heads = a.num_kv_heads; da.page_size = a.page_size; da.stride_page = a.page_stride;
  da.scale_log2 = a.sm_scale * 1.4426950408889634f;
  const int Hkv = a.num_kv_heads, q_dim = a.num_qo_heads * D, kv_dim = Hkv * D;
  const int sub = c.lane & 15, grp = c.lane >> 4;
  const __amdgpu_buffer_rsrc_t qr = act_rsrc(a.qkv_out, (q_dim + 2 * kv_dim) * 2);
  const int n_items = a.num_slots * Hkv;
  // scan waves per chunk exactly as launch_decode() picks them for the same plan (attn_decode.hip: 8 waves when the
  // launch has at most one workgroup per CU, else 4) - the token -> state partition decides the summation order
  const int nwv = n_items <= 256 ? 8 : 4;
  for (int item = c.wg; item < n_items; item += c.nwg) {
    const int slot = item / Hkv, kvh = item - slot * Hkv;
    const u32x4 d0 = *reinterpret_cast<const u32x4*>(a.slot_desc + 8 * slot);
    const u32x4 d1 = *reinterpret_cast<const u32x4*>(a.slot_desc + 8 * slot + 4);
    ChunkInfo ci;
    ci.b = (int)d0.x; ci.lo = (int)d0.y; ci.hi = (int)d0.z; ci.pbase = (int)d0.w;
    const int pos = (int)d1.x;
    ci.kv_len = (int)d1.y;
    const int s0 = (int)d1.z, s1 = (int)d1.w;
    if (ci.lo < 0) continue;  // padding slot (uniform for the workgroup)
    const Half* crow = a.cos_cache + (size_t)pos * D;
    const Half* srow = a.sin_cache + (size_t)pos * D;
    const int row_off = ci.b * (q_dim + 2 * kv_dim);
    u32x4 qv[GROUP];
#pragma unroll
    for (int h = 0; h < GROUP; ++h) {
      const u32x4 x = ld_act16(qr, (row_off + (kvh * GROUP + h) * D + sub * 8) * 2);
      qv[h] = head_norm_rope16(x, q_norm_w, crow, srow, sub, a.rms_eps);
    }
    const bool owns_new = pos >= ci.lo && pos < ci.hi;  // workgroup-uniform
    if (owns_new) {
      if (c.cw == 0 && grp == 0) {
        const u32x4 xk = ld_act16(qr, (row_off + q_dim + kvh * D + sub * 8) * 2);
        const u32x4 kn = head_norm_rope16(xk, k_norm_w, crow, srow, sub, a.rms_eps);
        const u32x4 xv = ld_act16(qr, (row_off + q_dim + kv_dim + kvh * D + sub * 8) * 2);
        const int page = a.page_indices[ci.pbase + pos / a.page_size];
        const long base = (long)page * a.page_stride + ((long)(pos % a.page_size) * Hkv + kvh) * D + sub * 8;
        *reinterpret_cast<u32x4*>(a.kv_data + base + da.k_off) = kn;
        *reinterpret_cast<u32x4*>(a.kv_data + base + da.v_off) = xv;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!eng_cbar(c)) return false;  // the new row is visible to the scanning waves of this workgroup
    }
    // the virtual scan waves of the plan over the 3 consumer waves: vw = cw, cw + 3, cw + 6
    for (int vw = c.cw; vw < nwv; vw += 3) {
      Scan st;
      st.init();
      st.scan(da, ci, qv, kvh, c.lane, vw, nwv);
      st.store_state(sm_m, sm_l, sm_o, vw * Scan::TPI + grp, c.lane);
    }
    if (!eng_cbar(c)) return false;
    const bool split = a.use_split != 0;
    if (c.cw == 0) {
      if (split) attn_finish_part<D, GROUP, true>(da, ci.b, slot, kvh, c.lane, nwv * Scan::TPI, sm_m, sm_l, sm_o, true);
      else attn_finish_part<D, GROUP, false>(da, ci.b, slot, kvh, c.lane, nwv * Scan::TPI, sm_m, sm_l, sm_o, false, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (split) {
        if (c.lane == 0) {
          int* ctr = a.merge_counters + (size_t)(ci.b * Hkv + kvh) * kMergeCtrStride;
          const int last = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s1 - s0 - 1;
          if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          lds_store_rel(c.sy + kSyFlag, (unsigned)last);
        }
      } else {
        eng_arrive(c, layer, kOpAttn, 0);
      }
    }
    if (split) {
      if (!eng_cbar(c)) return false;
      const bool last = lds_load_acq(c.sy + kSyFlag) != 0u;
      if (last) {
        for (int h = c.cw; h < GROUP; h += 3) {
          const int head = kvh * GROUP + h;
          merge_one<D, true, true>(a.tmp_v, a.tmp_s, s0, s1, head, a.num_qo_heads,
                                   a.attn_out + ((size_t)ci.b * a.num_qo_heads + head) * D);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!eng_cbar(c)) return false;   // also keeps the scratch / flag word stable until everybody has read them
      if (last && c.cw == 0) eng_arrive(c, layer, kOpAttn, 0);
    } else {
      if (!eng_cbar(c)) return false;
    }
  }
  return true;
}

template <int GROUP>
__device__ __forceinline__ void eng_consumer(EngCtx& c) {
  const EngArgs& a = *c.a;
  const EngShape s = eng_shape(a, c.nwg);
  u32x4* xs = (u32x4*)reinterpret_cast<lds_v4*>(c.lds + kRingBytes);
  float* scratch = (float*)reinterpret_cast<__attribute__((address_space(3))) float*>(c.lds + kRingBytes);
  uint16_t* outs = (uint16_t*)reinterpret_cast<__attribute__((address_space(3))) uint16_t*>(c.lds + kRingBytes + kScratchBytes);
  const unsigned per_shard = (unsigned)(c.nwg / kEngShards);
  const bool ks4_o = s.QD >= 4096, ks4_dn = s.I >= 4096, ks4_h = s.H >= 4096;
  const int bs = 1;
  unsigned long long tr[18], t_last = eng_now();
  const unsigned long long t_begin = t_last;
#pragma unroll
  for (int i = 0; i < 18; ++i) tr[i] = 0ull;
#define ENG_TR(k) do { if (a.trace) { const unsigned long long now_ = eng_now(); tr[k] += now_ - t_last; t_last = now_; } } while (0)
  const Half* cur = a.embed + (size_t)a.token_id[0] * s.H;   // layer 0: the embedding row is the residual stream
  bool cur_coherent = false;
  for (int layer = 0; layer < a.layers; ++layer) {
    const Half* ln1 = reinterpret_cast<const Half*>(eng_tab_ptr(c, layer, 4));
    const Half* ln2 = reinterpret_cast<const Half*>(eng_tab_ptr(c, layer, 5));
    const unsigned gl = (unsigned)layer * (unsigned)s.per_layer;
    const unsigned g_qkv = gl, g_o = gl + s.np0, g_gu = g_o + s.np1, g_dn = g_gu + s.np2, g_next = g_dn + s.np3;
    // ---- QKV: x = rms_norm(cur [+ mlp_out of the previous layer]) ----
    if (layer > 0) {
      if (!eng_wait_op(c, layer - 1, kOpDown, per_shard, 0)) return;
    ENG_TR(0);
      eng_stage_norm(c, xs, cur, cur_coherent, a.mlp_out, ln1, a.hidden_b, s.H, a.rms_eps);
      cur = a.hidden_b;
      cur_coherent = true;
    } else {
      eng_stage_norm(c, xs, cur, false, nullptr, ln1, nullptr, s.H, a.rms_eps);
    }
    if (!eng_cbar(c)) return;
    ENG_TR(1);
    if (ks4_h ? !eng_gemv_rows<true, false>(c, xs, outs, g_qkv, s.rows0, s.ppr0, g_o)
              : !eng_gemv_rows<false, false>(c, xs, outs, g_qkv, s.rows0, s.ppr0, g_o)) return;
    ENG_TR(2);
    if (!eng_store_out(c, outs, a.qkv_out, s.rows0, layer, kOpQkv)) return;
    ENG_TR(3);
    // ---- attention ----
    if (!eng_wait_op(c, layer, kOpQkv, per_shard, 0)) return;
    ENG_TR(4);
    if (!eng_attention<GROUP>(c, layer, scratch)) return;
    ENG_TR(5);
    // ---- O ----
    if (!eng_wait_op(c, layer, kOpAttn, 0, (unsigned)(a.num_kv_heads * bs))) return;
    ENG_TR(6);
    eng_stage_plain(c, xs, a.attn_out, s.QD);
    if (!eng_cbar(c)) return;
    ENG_TR(7);
    if (ks4_o ? !eng_gemv_rows<true, false>(c, xs, outs, g_o, s.rows1, s.ppr1, g_gu)
              : !eng_gemv_rows<false, false>(c, xs, outs, g_o, s.rows1, s.ppr1, g_gu)) return;
    ENG_TR(8);
    if (!eng_store_out(c, outs, a.attn_proj, s.rows1, layer, kOpO)) return;
    ENG_TR(9);
    // ---- gate|up: x = rms_norm(cur + attn_proj), SwiGLU epilogue ----
    if (!eng_wait_op(c, layer, kOpO, per_shard, 0)) return;
    ENG_TR(10);
    eng_stage_norm(c, xs, cur, cur_coherent, a.attn_proj, ln2, a.hidden_a, s.H, a.rms_eps);
    cur = a.hidden_a;
    cur_coherent = true;
    if (!eng_cbar(c)) return;
    ENG_TR(11);
    if (ks4_h ? !eng_gemv_rows<true, true>(c, xs, outs, g_gu, s.rows2 >> 1, s.ppr2, g_dn)
              : !eng_gemv_rows<false, true>(c, xs, outs, g_gu, s.rows2 >> 1, s.ppr2, g_dn)) return;
    ENG_TR(12);
    if (!eng_store_out(c, outs, a.act, s.rows2 >> 1, layer, kOpGu)) return;
    ENG_TR(13);
    // ---- down ----
    if (!eng_wait_op(c, layer, kOpGu, per_shard, 0)) return;
    ENG_TR(14);
    eng_stage_plain(c, xs, a.act, s.I);
    if (!eng_cbar(c)) return;
    ENG_TR(15);
    if (ks4_dn ? !eng_gemv_rows<true, false>(c, xs, outs, g_dn, s.rows3, s.ppr3, g_next)
               : !eng_gemv_rows<false, false>(c, xs, outs, g_dn, s.rows3, s.ppr3, g_next)) return;
    ENG_TR(16);
    if (!eng_store_out(c, outs, a.mlp_out, s.rows3, layer, kOpDown)) return;
    ENG_TR(17);
  }
  if (a.trace && c.cw == 0 && c.lane == 0) {
#pragma unroll
    for (int i = 0; i < 18; ++i) a.trace[(size_t)c.wg * 32 + i] = tr[i];
    a.trace[(size_t)c.wg * 32 + 18] = c.t_wait;
    a.trace[(size_t)c.wg * 32 + 22] = eng_now() - t_begin;
  }
#undef ENG_TR
}

template <int GROUP>
__global__ __launch_bounds__(256) void decode_engine_kernel(const EngArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char eng_lds[];
  EngCtx c;
  c.a = &a;
  c.lds = (lds_u8*)eng_lds;
  c.sy = reinterpret_cast<lds_u32*>(c.lds + kRingBytes + kScratchBytes + kOutsBytes);
  c.ltab = c.lds + kRingBytes + kScratchBytes + kOutsBytes + kSyncBytes;
  {  // layer table -> LDS (16 dwords per layer): the loader must not issue a single counted VMEM load later on
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.layer_table);
    lds_u32* dst = reinterpret_cast<lds_u32*>(c.lds + kRingBytes + kScratchBytes + kOutsBytes + kSyncBytes);
    for (int i = threadIdx.x; i < a.layers * 16; i += 256) dst[i] = src[i];
  }
  c.wg = blockIdx.x; c.nwg = gridDim.x; c.lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.cw = wave - 1;
  c.bar_gen = 0; c.ready_seen = 0; c.dead = false; c.t_wait = 0ull;
  if (threadIdx.x < 16) c.sy[threadIdx.x] = 0u;
  __syncthreads();
  if (wave == 0) eng_loader(c);
  else eng_consumer<GROUP>(c);
}

}  // namespace pk

using namespace pk;

extern "C" {

int32_t pegainfer_decode_engine_lds_bytes(void) { return kEngLdsBytes; }

// 0 = the model shape / device can run the engine with `num_workgroups` workgroups (one per CU)
int32_t pegainfer_decode_engine_supported(const pegainfer_engine_args_t* a, int32_t num_workgroups) {
  if (!a || num_workgroups < kEngShards || num_workgroups % kEngShards || a->layers < 1 || a->layers > kEngMaxLayers) return -1;
  const int H = a->hidden, I = a->intermediate, QD = a->num_qo_heads * a->head_dim, KVD = a->num_kv_heads * a->head_dim;
  if (a->head_dim != 128 || a->num_kv_heads <= 0 || a->num_qo_heads != 4 * a->num_kv_heads) return -1;
  if (H % 512 || I % 512 || QD % 512 || H > 4096 || I * 2 > kScratchBytes || QD * 2 > kScratchBytes) return -1;
  const int nw = num_workgroups;
  const int rows[4] = {QD + 2 * KVD, H, I, H};
  for (int r : rows)
    if (r % nw || ((r / nw) & 1) || r / nw > 256) return -1;
  return 0;
}

int32_t pegainfer_decode_engine_step(const pegainfer_engine_args_t* a, int32_t num_workgroups, pegainfer_stream_t stream) {
  if (pegainfer_decode_engine_supported(a, num_workgroups)) return static_cast<int32_t>(hipErrorInvalidValue);
  if (a->num_slots <= 0 || !host_aligned16(a->slot_desc) || !host_aligned16(a->qkv_out) || !host_aligned16(a->kv_data))
    return static_cast<int32_t>(hipErrorInvalidValue);
  auto kern = &decode_engine_kernel<4>;
  static const bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kEngLdsBytes), true);
  (void)once;
  kern<<<num_workgroups, 256, kEngLdsBytes, as_stream(stream)>>>(*a);
  return static_cast<int32_t>(hipGetLastError());
}

}  // extern "C"
