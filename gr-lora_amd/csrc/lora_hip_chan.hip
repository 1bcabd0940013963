// lora_hip_chan.hip -- gfx950 channeliser (frequency shift + polyphase arbitrary resampler) and its C ABI
// (include/lora_hip_chan.h).  Product code: never links or calls anything under oracle/.
//
// What it replaces: blocks_rotator_cc and pfb_arb_resampler_ccf in front of lora.demod in the reference's
// receive flowgraph (examples/rx_usrp.grc:489-503, :825-860).  One pass over the samples in HBM:
//   * the rotation is applied while a tile is loaded into LDS, from the exact phase (a double per tile, two
//     small tables inside it) instead of the reference's sample-by-sample float product chain, which cannot
//     run in parallel and whose rounding depends on the VOLK kernel picked at run time;
//   * whole-number decimation D (d_flt_rate == 0 and d_dec_rate = D*nfilts: 1 MS/s -> 250 kS/s in the
//     flowgraph) uses one arm for every output, so the filter bank is a plain decimating FIR:
//     chan_decim_kernel splits it into its D polyphase components, one lane per component, 8 consecutive
//     outputs per lane in registers, every LDS value used for up to 8 multiply-adds, then a DPP sum over
//     the lanes of a group;
//   * any other rate runs chan_arb_kernel (filter banks in LDS) or chan_general_kernel (banks too big for LDS):
//     one output per lane, arm and interpolation weight per output from the closed form of the reference's
//     recurrence, taps applied in the reference's order with separately rounded products, so both are
//     bit-identical to a sequential float evaluation when the rotation is off.
#include "lora_hip_chan.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <utility>
#include <vector>

namespace lora_chan {

constexpr int kBlock = 256;
constexpr int kMaxSteps = 16;          // tile loads per thread the rotation table covers
constexpr int kFracBits = 24;          // d_acc as an integer count of 2^-24

struct CParams {
  const float2 *in;                    // this call's samples
  const float2 *carry;                 // the K-1 samples before in[0]
  long long n_in;
  float2 *out;
  unsigned long long n_out;
  const float *filt, *dfilt;           // [nfilts][K], stored reversed as fir_filter::set_taps does
  const float2 *bank2;                 // chan_arm_kernel: the two banks interleaved, {f0, f1}[nfilts][K]
  unsigned tile_cap;                   // chan_arm_kernel: float2 slots of the LDS tile (the order list lies behind it)
  unsigned row_pitch;                  // chan_arm_kernel: slots per row of the de-interleaved tile (sample i sits in row i mod 2^E)
  unsigned K, nfilts;
  // closed form of the reference's index recurrence (exact, see chan_work_chunk): output m sits at polyphase
  // position p0 + m*dec + ((a0 + m*a) >> 24) relative to in[0], with d_acc = ((a0 + m*a) mod 2^24) / 2^24
  unsigned long long p0, dec, a0, a;
  const unsigned *sched_pos;           // fallback form: position per output ...
  const float *sched_acc;              // ... and d_acc per output, from the host
  unsigned tile_out;                   // general kernel: outputs per tile
  long long i0;                        // decim kernel: input index of output 0
  unsigned jfix;                       // decim kernel: the one arm in use
  double phase0, inc;                  // rotation: angle of sample s (relative to in[0]) = phase0 + inc*s
  double step_re, step_im;             // decim kernel: exp(i*inc*D*TOB*gridDim.x)
  const float2 *rot_lane;              // decim kernel: exp(i*inc*t), t < 256
  float2 rot_step[kMaxSteps];          // decim kernel: exp(i*inc*256*k)
};

// sample s of the stream (history ++ input), relative to in[0]; zero outside what exists.  Branch-free: the
// address is clamped to something valid (the host launches only with n_in >= 1) and the value selected.
__device__ __forceinline__ float2 fetch(const CParams &p, const long long s) {
  const long long c = s + (long long)(p.K - 1);
  const bool in_main = s >= 0 && s < p.n_in, in_carry = s < 0 && c >= 0;
  const float2 *ptr = in_main ? p.in + s : (in_carry ? p.carry + c : p.in);
  const float2 v = *ptr;
  return (in_main || in_carry) ? v : make_float2(0.0f, 0.0f);
}
// rotator::rotate: z = in * d_phase, separately rounded products
__device__ __forceinline__ float2 rotate(const float2 x, const float2 ph) {
  return make_float2(x.x * ph.x - x.y * ph.y, x.x * ph.y + x.y * ph.x);
}
__device__ __forceinline__ float2 cmul_f(const float2 a, const float2 b) {
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// ---- any rate ------------------------------------------------------------------------------------------
// pfb_arb_resampler_ccf::filter, one output per lane.  A tile of outputs shares the span of input samples it
// touches through LDS (read once from HBM, rotated on the way in).
// MODE 0: closed form, d_flt_rate == 0 (one weight-free arm per output); 1: closed form with interpolation;
// 2: positions and weights from the host (rates above nfilts, where the float recurrence is not exact).
template <bool ROT, int MODE>
__global__ __launch_bounds__(kBlock) void chan_general_kernel(const CParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *tile = reinterpret_cast<float2 *>(smem_raw);
  const int tid = threadIdx.x;
  const unsigned K = p.K;
  constexpr unsigned long long kMask = (1ull << kFracBits) - 1;
  auto pos = [&](const unsigned long long m) -> unsigned long long {
    if (MODE == 2) return p.sched_pos[m];
    return p.p0 + m * p.dec + ((p.a0 + m * p.a) >> kFracBits);
  };
  const unsigned long long n_tiles = (p.n_out + p.tile_out - 1) / p.tile_out;
  for (unsigned long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const unsigned long long m0 = t * p.tile_out;
    const unsigned long long m1 = (m0 + p.tile_out < p.n_out) ? m0 + p.tile_out : p.n_out;
    const long long s_lo = (long long)(pos(m0) / p.nfilts) - (long long)(K - 1);
    const int span = (int)((long long)(pos(m1 - 1) / p.nfilts) - s_lo + 1);   // host guarantees it fits the tile
    for (int i = tid; i < span; i += kBlock) {
      float2 x = fetch(p, s_lo + i);
      if (ROT) {
        double sn, cs;
        sincos(p.phase0 + p.inc * (double)(s_lo + i), &sn, &cs);
        x = rotate(x, make_float2((float)cs, (float)sn));
      }
      tile[i] = x;
    }
    __syncthreads();
    const unsigned long long m = m0 + tid;
    if (m < m1) {
      const unsigned long long P = pos(m);
      const unsigned j = (unsigned)(P % p.nfilts);
      const float2 *x = tile + ((long long)(P / p.nfilts) - (long long)(K - 1) - s_lo);
      const float *__restrict__ f0 = p.filt + (size_t)j * K;
      const float *__restrict__ f1 = p.dfilt + (size_t)j * K;
      // volk_32fc_32f_dot_prod_32fc (generic): sequential float accumulation, product rounded before the add
      float re0 = 0, im0 = 0, re1 = 0, im1 = 0;
      for (unsigned k = 0; k < K; k++) {
        const float2 v = x[k];
        const float t0 = f0[k];
        re0 += v.x * t0; im0 += v.y * t0;
        if (MODE != 0) {
          const float t1 = f1[k];
          re1 += v.x * t1; im1 += v.y * t1;
        }
      }
      float2 o = make_float2(re0, im0);
      if (MODE != 0) {                   // o0 + o1*d_acc; with d_flt_rate == 0 d_acc stays 0 and the term is dropped
        const float a = (MODE == 2) ? p.sched_acc[m]
                                    : (float)((p.a0 + m * p.a) & kMask) * (1.0f / (float)(1u << kFracBits));
        o.x = re0 + re1 * a; o.y = im0 + im1 * a;
      }
      p.out[m] = o;
    }
    __syncthreads();
  }
}

// ---- any rate whose filter banks fit LDS ------------------------------------------------------------------
// The same sums as chan_general_kernel in the same order (separately rounded products, sequential float
// accumulation, o0 + o1*d_acc), so the result is bit-identical to it and to the oracle's float evaluation; the
// difference is where the taps come from: the arm bank and the difference bank sit in LDS interleaved as
// {f0, f1}[arm][k] (row stride odd, so lanes on different arms spread over the banks) instead of being gathered
// from global memory per lane.  Two 8-byte LDS reads per tap and output and nothing shared between lanes (every
// lane has its own arm and its own window): LDS-bandwidth-bound by construction.
// INTERP false: d_flt_rate == 0 (d_acc stays 0, the difference filter is not evaluated, as in mode 0 there).
template <bool ROT, bool INTERP>
__global__ __launch_bounds__(kBlock) void chan_arb_kernel(const CParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned K = p.K, KP = K | 1u;
  float2 *bank = reinterpret_cast<float2 *>(smem_raw);          // [nfilts][KP]
  float2 *tile = bank + (size_t)p.nfilts * KP;
  const int tid = threadIdx.x;
  constexpr unsigned long long kMask = (1ull << kFracBits) - 1;
  for (unsigned i = tid; i < p.nfilts * K; i += kBlock) {
    const unsigned j = i / K, k = i - j * K;
    bank[j * KP + k] = make_float2(p.filt[i], p.dfilt[i]);
  }
  auto pos = [&](const unsigned long long m) -> unsigned long long {
    return p.p0 + m * p.dec + ((p.a0 + m * p.a) >> kFracBits);
  };
  const unsigned long long n_tiles = (p.n_out + p.tile_out - 1) / p.tile_out;
  for (unsigned long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const unsigned long long m0 = t * p.tile_out;
    const unsigned long long m1 = (m0 + p.tile_out < p.n_out) ? m0 + p.tile_out : p.n_out;
    const long long s_lo = (long long)(pos(m0) / p.nfilts) - (long long)(K - 1);
    const int span = (int)((long long)(pos(m1 - 1) / p.nfilts) - s_lo + 1);
    __syncthreads();                                             // the previous tile's readers are done (and the bank is visible)
    for (int i = tid; i < span; i += kBlock) {
      float2 x = fetch(p, s_lo + i);
      if (ROT) {
        double sn, cs;
        sincos(p.phase0 + p.inc * (double)(s_lo + i), &sn, &cs);
        x = rotate(x, make_float2((float)cs, (float)sn));
      }
      tile[i] = x;
    }
    __syncthreads();
    const unsigned long long m = m0 + tid;
    if (m < m1) {
      const unsigned long long P = pos(m);
      const unsigned j = (unsigned)(P % p.nfilts);
      const float2 *__restrict__ x = tile + ((long long)(P / p.nfilts) - (long long)(K - 1) - s_lo);
      const float2 *__restrict__ f = bank + j * KP;
      // volk_32fc_32f_dot_prod_32fc (generic) twice: sequential float accumulation, product rounded before the add
      float re0 = 0, im0 = 0, re1 = 0, im1 = 0;
#pragma unroll 4
      for (unsigned k = 0; k < K; k++) {
        const float2 tp = f[k], v = x[k];
        re0 += v.x * tp.x; im0 += v.y * tp.x;
        if (INTERP) { re1 += v.x * tp.y; im1 += v.y * tp.y; }
      }
      float2 o = make_float2(re0, im0);
      if (INTERP) {                                              // o0 + o1*d_acc
        const float a = (float)((p.a0 + m * p.a) & kMask) * (1.0f / (float)(1u << kFracBits));
        o.x = re0 + re1 * a; o.y = im0 + im1 * a;
      }
      p.out[m] = o;
    }
  }
}


// ---- any rate, outputs grouped by arm (nfilts <= kArmMaxFilts) -------------------------------------------------
// chan_arb_kernel gives every lane its own arm, so every tap is a per-lane LDS read beside the sample's: two LDS reads
// per tap and output, and the LDS -- one pipe for the CU's four SIMDs -- binds (measured: 57 % busy, a third of it bank
// conflicts).  Here a tile's outputs are first sorted by arm (a counting sort in LDS: one pass counts, one pass
// scatters their numbers into `order`, each arm's run padded to whole waves), and a wave then works on 64 outputs of
// ONE arm: the tap pair {f0[k], f1[k]} is the same for all its lanes and comes through the scalar data path (constant
// address space, s_load into SGPRs -- reads only), the banks need no LDS at all (the tile of samples is about three
// times as big instead), and what is left per tap and lane is one LDS read and the eight multiplies and adds of the
// reference's sums -- same products, same order (sequential accumulation, product rounded before the add, o0 + o1*d_acc),
// so the result is still bit-identical to the oracle's float evaluation.  Which lane computes which output has no
// influence on an output's value.
#ifndef LORA_ARM_BLOCK
#define LORA_ARM_BLOCK 512
#endif
#ifndef LORA_ARM_WG_PER_CU
#define LORA_ARM_WG_PER_CU 2
#endif
#ifndef LORA_ARM_ABL
#define LORA_ARM_ABL 0   // timing-only ablations (wrong results): 1 taps are constants (no scalar loads), 2 samples are constants (no LDS
                         // reads in the filter loop), 4 no tile load from HBM, 8 no filter loop (load + sort + store only), 16 no sort (identity order)
#endif
constexpr int kArmBlock = LORA_ARM_BLOCK;          // threads per workgroup
constexpr int kArmWgPerCu = LORA_ARM_WG_PER_CU;    // resident workgroups per CU: independent tiles in different phases (loading, sorting, filtering)
constexpr unsigned kArmMaxFilts = 64;
typedef const __attribute__((address_space(4))) float *const_f_ptr;     // constant address space: uniform loads are scalar loads
// one ds_read_b64 per sample: paired into ds_read2_b64 the same bytes take twice the LDS cycles (MI355X_MICROARCH.md, LDS
// table), and the LDS is the one pipe all four SIMDs share; a volatile LDS access is never paired
// (addressed by a 32-bit LDS byte address: through generic pointers every read drags a null check and 64-bit arithmetic along)
__device__ __forceinline__ float2 lds_ld(const unsigned lds_byte_addr) {
  if (LORA_ARM_ABL & 2) return make_float2((float)lds_byte_addr, 1.0f);
  typedef float lds_v2f __attribute__((ext_vector_type(2)));
  const lds_v2f t = *(const volatile __attribute__((address_space(3))) lds_v2f *)(size_t)lds_byte_addr;
  return make_float2(t.x, t.y);
}
// E: the tile is stored de-interleaved into 2^E rows, sample i (relative to the tile's first) in row i mod 2^E at slot i >> E.
// Consecutive outputs of one arm start a near-constant number S of inputs apart; the lanes of a wave then read samples
// S apart, and an even S puts 32 lanes on 16 (S = 2 mod 4) or 8 bank pairs: every ds_read_b64 takes twice or four times its
// cycles, which makes the LDS, not the multiplies, the bound (measured at rate 0.3, S = 10: 131 M conflict cycles of 258 M).
// With 2^E the power of two in S all lanes stay in one row and S >> E apart in it -- odd, so 32 lanes hit 32 different
// bank pairs.  The host picks E from the rate (arm_deinterleave); any E gives the same results.
template <bool ROT, bool INTERP, int E>
__global__ __launch_bounds__(kArmBlock, (kArmBlock / 64) * kArmWgPerCu / 4) void chan_arm_kernel(const CParams p) {
  constexpr unsigned NR = 1u << E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *tile = reinterpret_cast<float2 *>(smem_raw);                             // [tile_cap]
  const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;   // its LDS byte address
  unsigned short *order = reinterpret_cast<unsigned short *>(tile + p.tile_cap);   // [tile_out + 63*nfilts]: runs of one arm
  unsigned short *xoff = order + (p.tile_out + 63u * p.nfilts);                    // [tile_out]: where an output's window starts in the tile
  unsigned char *arm_of_out = reinterpret_cast<unsigned char *>(xoff + p.tile_out);   // [tile_out]
  constexpr int kWaves = kArmBlock / 64;
  __shared__ unsigned cntw[kWaves][kArmMaxFilts];   // outputs per (wave's share of the tile, arm); then each wave's write cursors
  __shared__ unsigned n_chunks;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned K = p.K, nf = p.nfilts;
  constexpr unsigned long long kMask = (1ull << kFracBits) - 1;
  auto pos = [&](const unsigned long long m) -> unsigned long long {
    return p.p0 + m * p.dec + ((p.a0 + m * p.a) >> kFracBits);
  };
  // position -> (input index, arm); the filter count is a power of two in every flowgraph (32), which spares the 64-bit division
  const bool pow2 = (nf & (nf - 1)) == 0;
  const unsigned sh = 31u - (unsigned)__builtin_clz(nf);
  auto arm_of = [&](const unsigned long long P) -> unsigned { return pow2 ? ((unsigned)P & (nf - 1)) : (unsigned)(P % nf); };
  auto in_of = [&](const unsigned long long P) -> long long { return (long long)(pow2 ? (P >> sh) : (P / nf)); };
  const unsigned long long n_tiles = (p.n_out + p.tile_out - 1) / p.tile_out;
  const unsigned H = p.row_pitch;
  auto slot = [&](const unsigned i) -> unsigned { return (i & (NR - 1)) * H + (i >> E); };      // sample i of the tile -> LDS slot
  for (int i = tid; i < kWaves * (int)kArmMaxFilts; i += kArmBlock) (&cntw[0][0])[i] = 0;
  for (unsigned long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const unsigned long long m0 = t * p.tile_out;
    const unsigned long long m1 = (m0 + p.tile_out < p.n_out) ? m0 + p.tile_out : p.n_out;
    const unsigned nt = (unsigned)(m1 - m0);
    // the tile's first output: everything about output m0 + i follows from these in a few 32x32 -> 64 bit products
    const unsigned long long P0 = pos(m0), A0 = (p.a0 + m0 * p.a) & kMask;
    const unsigned arm0 = arm_of(P0);
    const long long in0 = in_of(P0), s_lo = in0 - (long long)(K - 1);
    const int span = (int)(in_of(pos(m1 - 1)) - s_lo + 1);
    const unsigned dec32 = (unsigned)p.dec, a32 = (unsigned)p.a;    // both below 2^32 (the host checks dec; a < 2^24): 32 x 32 -> 64 bit products
    auto rel_of = [&](const unsigned i) -> unsigned long long {       // pos(m0 + i) - P0 + arm0
      return (unsigned long long)arm0 + (unsigned long long)i * dec32 + ((A0 + (unsigned long long)i * a32) >> kFracBits);
    };
    __syncthreads();                                             // the previous tile's readers are done; the counters are zero
    // Tile load.  All but the stream's first and last tiles lie wholly inside this call's input: then the loads need no
    // selection between carry, input and zero, and go out eight per thread before the first is waited for (as one load,
    // one wait, one store per sample -- what the general form compiles to -- a tile paid fourteen HBM round trips).
    auto put = [&](const int i, float2 x) {
      if (ROT) {
        // the exact phase of sample s (a double), reduced to (-pi, pi], then a float sine and cosine: the phasor is
        // within 1e-7 of the exact one (the reference's own product chain wanders 8e-6 around it)
        const double a = p.phase0 + p.inc * (double)(s_lo + i);
        const float r = (float)(a - 6.283185307179586476925 * rint(a * 0.15915494309189533577));
        float sn, cs;
        sincosf(r, &sn, &cs);
        x = rotate(x, make_float2(cs, sn));
      }
      tile[slot((unsigned)i)] = x;
    };
    if (!(LORA_ARM_ABL & 4) && s_lo >= 0 && s_lo + span <= p.n_in) {
      const float2 *__restrict__ src = p.in + s_lo;
      constexpr int U = 8;
      for (int i0 = tid; i0 < span; i0 += kArmBlock * U) {
        float2 tmp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int i = i0 + u * kArmBlock;
          tmp[u] = src[i < span ? i : span - 1];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int i = i0 + u * kArmBlock;
          if (i < span) put(i, tmp[u]);
        }
      }
    } else {
      for (int i = tid; i < span; i += kArmBlock) put(i, (LORA_ARM_ABL & 4) ? make_float2((float)i, 1.0f) : fetch(p, s_lo + i));
    }
    // Sort the tile's outputs by arm, keeping their order inside an arm (consecutive outputs of one arm have window
    // starts a near-constant stride apart, which spreads a wave's LDS reads over the banks; a random order does not).
    // Wave w owns the outputs [w*per, (w+1)*per) of the tile.  Pass 1: arm and window start of every output, and how
    // many outputs of each arm the wave's share holds.
    const unsigned per = (nt + kWaves - 1) / kWaves;
    const unsigned w_lo = wave * per, w_hi = (w_lo + per < nt) ? w_lo + per : nt;
    for (unsigned i = w_lo + lane; i < w_hi; i += 64) {
      const unsigned long long rel = rel_of(i);
      const unsigned a = pow2 ? ((unsigned)rel & (nf - 1)) : (unsigned)(rel % nf);
      const unsigned d_in = (unsigned)(pow2 ? (rel >> sh) : (rel / nf));          // input index relative to in0
      arm_of_out[i] = (unsigned char)a;
      xoff[i] = (unsigned short)d_in;                            // window start = in0 + d_in - (K-1) - s_lo = d_in
      atomicAdd(&cntw[wave][a], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane = arm: its run starts behind the (padded) runs of the arms before it (an exclusive scan over the lanes of
      // this one wave); inside the run the waves' shares follow each other.  cntw[w][arm] becomes wave w's write cursor.
      unsigned n_w[kWaves], tot = 0;
      const bool is_arm = lane < (int)nf;
#pragma unroll
      for (int w = 0; w < kWaves; w++) { n_w[w] = is_arm ? cntw[w][lane] : 0u; tot += n_w[w]; }
      const unsigned padded = (tot + 63u) & ~63u;
      unsigned incl = padded;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const unsigned off = incl - padded;
      if (is_arm) {
        unsigned c = off;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { cntw[w][lane] = c; c += n_w[w]; }
        for (unsigned s2 = off + tot; s2 < off + padded; s2++) order[s2] = 0xFFFFu;     // the padding computes nothing
        if (lane == (int)nf - 1) n_chunks = (off + padded) >> 6;
      }
    }
    __syncthreads();
    // Pass 2: every output takes the next slot behind its wave's cursor for its arm.  A wave's LDS atomics execute in
    // program order and, within one instruction, lane after lane, so a wave's share comes out in output order (if some
    // hardware did not, only the bank spread of the filter loop's reads would change, not a result).
    for (unsigned i = w_lo + lane; i < w_hi; i += 64) order[atomicAdd(&cntw[wave][arm_of_out[i]], 1u)] = (unsigned short)i;
    __syncthreads();
    for (int i = tid; i < kWaves * (int)kArmMaxFilts; i += kArmBlock) (&cntw[0][0])[i] = 0;     // the cursors are spent: zero for the next tile
    const unsigned nc = n_chunks;
    for (unsigned c = wave; c < nc; c += kWaves) {
      const unsigned e = order[c * 64 + lane];
      const bool valid = e != 0xFFFFu;
      const unsigned e0 = __builtin_amdgcn_readfirstlane(e);     // a run starts on a wave boundary: lane 0 is never padding
      const unsigned i = valid ? e : e0;
      const unsigned arm = __builtin_amdgcn_readfirstlane((unsigned)arm_of_out[i]);   // the same for every lane of the chunk
      // sample base + NR*j + q of this lane's window sits at LDS byte address row[q] + 8*j
      const unsigned base = xoff[i];
      unsigned row[NR];
#pragma unroll
      for (unsigned q = 0; q < NR; q++) {
        const unsigned rq = (base + q) & (NR - 1);
        row[q] = tile_lds + 8u * (rq * H + ((base + q) >> E));
      }
      const const_f_ptr fs = (const_f_ptr)(p.bank2 + (size_t)arm * K);
      auto f = [&](const unsigned k2) {                                                                      // {f0[k], f1[k]}
        if (LORA_ARM_ABL & 1) return make_float2(0.5f + (float)k2, 0.25f);
        return make_float2(fs[2 * k2], fs[2 * k2 + 1]);
      };
      // volk_32fc_32f_dot_prod_32fc (generic) twice: sequential float accumulation, product rounded before the add.
      // Blocks of eight taps, the next block's taps (one s_load_dwordx16) and samples (eight ds_read_b64) requested
      // before the current block's 64 multiplies and adds: scalar loads return out of order, so every wait on them
      // is a wait for everything, and a request one block ahead has landed by the time it is waited for.
      float re0 = 0, im0 = 0, re1 = 0, im1 = 0;
      auto mac = [&](const float2 tp, const float2 v) {
        re0 += v.x * tp.x; im0 += v.y * tp.x;
        if (INTERP) { re1 += v.x * tp.y; im1 += v.y * tp.y; }
      };
      constexpr unsigned B = 8;
      auto load = [&](float2 (&tp)[B], float2 (&v)[B], const unsigned k0) {
#pragma unroll
        for (unsigned q = 0; q < B; q++) { tp[q] = f(k0 + q); v[q] = lds_ld(row[q & (NR - 1)] + 8u * (q >> E)); }
      };
      auto next_block = [&]() {
#pragma unroll
        for (unsigned q = 0; q < NR; q++) row[q] += 8u * (B >> E);
      };
      auto comp = [&](const float2 (&tp)[B], const float2 (&v)[B]) {
#pragma unroll
        for (unsigned q = 0; q < B; q++) mac(tp[q], v[q]);
      };
      unsigned k = 0;
      const unsigned nb = (LORA_ARM_ABL & 8) ? 0 : K / B;        // full blocks; two register sets take turns (no copies)
      if (nb) {
        float2 tA[B], vA[B], tB[B], vB[B];
        load(tA, vA, 0);
        unsigned bl = 1;
        for (; bl + 1 < nb; bl += 2) {
          next_block(); load(tB, vB, bl * B); comp(tA, vA);
          next_block(); load(tA, vA, (bl + 1) * B); comp(tB, vB);
        }
        if (bl < nb) { next_block(); load(tB, vB, bl * B); comp(tA, vA); comp(tB, vB); }
        else comp(tA, vA);
        k = nb * B;
      }
      if (k < K && !(LORA_ARM_ABL & 8)) {
        // the last K mod 8 taps as one more block, requested together (one wait instead of one per tap)
        const unsigned R = K - k;                                // wave-uniform
        float2 tp[B], v[B];
        if (k) {
#pragma unroll
          for (unsigned q = 0; q < NR; q++) row[q] += 8u * (B >> E);
        }
#pragma unroll
        for (unsigned q = 0; q < B; q++)
          if (q < R) { tp[q] = f(k + q); v[q] = lds_ld(row[q & (NR - 1)] + 8u * (q >> E)); }
#pragma unroll
        for (unsigned q = 0; q < B; q++)
          if (q < R) mac(tp[q], v[q]);
      }
      float2 o = make_float2(re0, im0);
      if (INTERP) {                                              // o0 + o1*d_acc
        const float a = (float)((A0 + (unsigned long long)i * a32) & kMask) * (1.0f / (float)(1u << kFracBits));
        o.x = re0 + re1 * a; o.y = im0 + im1 * a;
      }
      if (valid) p.out[m0 + i] = o;
    }
  }
}

// ---- whole-number decimation ---------------------------------------------------------------------------
// out[m] = sum_u g[u] * X[i0 + D*m - u], g = the arm's taps in natural order, u < K.  With u = D*q + ph:
//   out[m] = sum_ph sum_q g[D*q + ph] * Z_ph[m - q],   Z_ph[n] = X[i0 + D*n - ph]
// D stride-1 filters of Q = ceil(K/D) taps.  Lane (grp, ph) of a workgroup holds R = 8 consecutive outputs of
// component ph in registers; its taps stay in VGPRs for the whole launch; a window value read from LDS feeds
// up to 8 multiply-adds.  LDS holds the tile de-interleaved by component, L[ph][n], rows padded by one per 8
// and row length = 16/D (mod 16), which keeps both the de-interleaving writes and the window reads free of
// bank conflicts.  The next tile's samples are loaded into registers before the multiply-adds of the current
// one start, so that HBM latency hides behind them.
constexpr int decim_lanes(int d) { return d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : 16; }   // lanes per group: next power of two
template <int D, int NCH>
struct DecimGeom {
  // LG lanes form a group; lane ph < D owns component ph, the others (D not a power of two) hold zero taps
  static constexpr int LG = decim_lanes(D);
  static constexpr int Q8 = NCH * 8, R = 8, G = kBlock / LG, TOB = R * G;
  static constexpr int W = TOB + Q8 - 1;                               // window entries per component
  static constexpr int RL0 = W + (W >> 3) + 1;
  static constexpr int RL = RL0 + ((16 / LG - RL0 % 16) + 16) % 16;    // = 16/LG (mod 16)
  static constexpr int NI = (D * W + kBlock - 1) / kBlock;             // loads per thread per tile
  static_assert(D >= 2 && D <= 16, "whole-number decimation by 2..16");
  static_assert(NI <= kMaxSteps, "rotation step table too small");
  static_assert(RL % 16 == (16 / LG) % 16, "row length residue");
  static_assert((kBlock / LG) % 8 == 0 && R % 8 == 0, "pad split: thread part and constant part never carry");
};

template <int D, int NCH, bool ROT>
__global__ __launch_bounds__(kBlock, 4) void chan_decim_kernel(const CParams p) {
  using Gm = DecimGeom<D, NCH>;
  constexpr int Q8 = Gm::Q8, R = Gm::R, TOB = Gm::TOB, W = Gm::W, RL = Gm::RL, NI = Gm::NI, LG = Gm::LG;
  __shared__ float2 L[D * RL];
  const int tid = threadIdx.x;
  const int ph = tid % LG, grp = tid / LG;
  const bool owner = ph < D;                                           // lanes beyond D (D not a power of two) carry zero taps
  const unsigned K = p.K;
  // this lane's taps: g[D*q + ph] = reversed arm [K-1 - (D*q + ph)], zero beyond the arm
  float gp[Q8];
  {
    const float *__restrict__ arm = p.filt + (size_t)p.jfix * K;
#pragma unroll
    for (int q = 0; q < Q8; q++) {
      const unsigned u = (unsigned)(D * q + ph);
      gp[q] = (owner && u < K) ? arm[K - 1 - u] : 0.0f;
    }
  }
  // rotation: phase of the tile's first sample in double, advanced by one grid stride per iteration
  double br = 1.0, bi = 0.0;
  float2 lane_rot = make_float2(1.0f, 0.0f);
  const long long s_first = p.i0 - (long long)D * (Q8 - 1) - (D - 1);   // first sample of tile 0's span
  // Workgroups are dealt to the 8 XCDs round robin; neighbouring tiles share D*(Q8-1) samples of halo, so tile
  // numbers are handed out such that neighbours run on one XCD at about the same time and the halo is read from
  // HBM once, into that XCD's L2 (measured before: 13 % more HBM reads than input bytes).
  unsigned lb = blockIdx.x;
  if ((gridDim.x & 7u) == 0) lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (ROT) {
    sincos(p.phase0 + p.inc * (double)(s_first + (long long)lb * (D * TOB)), &bi, &br);
    lane_rot = p.rot_lane[tid];
  }
  const unsigned long long n_tiles = (p.n_out + TOB - 1) / TOB;
  // A tile whose span lies inside this call's input (all but the first and last few) loads unconditionally
  // from a wave-uniform base; the edges go through the guarded fetch.
  float2 xs[NI];
  auto load_tile = [&](const unsigned long long t) {
    const long long s_lo = s_first + (long long)t * (D * TOB);
    const bool interior = (s_lo >= 0) && (s_lo + (long long)(D * W) <= p.n_in);
    if (interior) {
      const float2 *__restrict__ src = p.in + s_lo;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        if (i < NI - 1 || tid + kBlock * i < D * W) xs[i] = src[tid + kBlock * i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; i++) xs[i] = fetch(p, s_lo + (tid + kBlock * i));
    }
  };
  unsigned long long t = lb;
  if (t < n_tiles) load_tile(t);
  while (t < n_tiles) {
    // ---- rotate, de-interleave: stream element e of the span goes to L[D-1 - e%D][e/D] -----------
    {
      float2 pt = make_float2(1.0f, 0.0f);
      if (ROT) pt = cmul_f(make_float2((float)br, (float)bi), lane_rot);
      if constexpr (LG == D) {
        float2 *row = L + (D - 1 - ph) * RL + grp + (grp >> 3);
        constexpr int GI = kBlock / D;                                 // e / D = grp + GI*i; GI*i is a multiple of 8
#pragma unroll
        for (int i = 0; i < NI; i++) {
          if (i < NI - 1 || tid + kBlock * i < D * W) {
            float2 x = xs[i];
            if (ROT) x = cmul_f(x, cmul_f(pt, p.rot_step[i]));           // FMA form: the phase is exact here, the product order is not the contract
            row[GI * i + ((GI * i) >> 3)] = x;
          }
        }
      } else {
        // D not a power of two: element e = tid + 256*i goes to component D-1 - e%D, entry e/D (division by a constant)
#pragma unroll
        for (int i = 0; i < NI; i++) {
          const int e = tid + kBlock * i;
          if (i < NI - 1 || e < D * W) {
            float2 x = xs[i];
            if (ROT) x = cmul_f(x, cmul_f(pt, p.rot_step[i]));
            const int a = e / D, b = e - a * D;
            L[(D - 1 - b) * RL + a + (a >> 3)] = x;
          }
        }
      }
      if (ROT) {                        // advance the tile phase by one grid stride (double complex product)
        const double nr = br * p.step_re - bi * p.step_im, ni = br * p.step_im + bi * p.step_re;
        br = nr; bi = ni;
      }
    }
    __syncthreads();
    const unsigned long long t_next = t + gridDim.x;
    if (t_next < n_tiles) load_tile(t_next);                           // in flight during the multiply-adds below
    // ---- R outputs x Q8 taps of this lane's component ---------------------------------------------
    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = make_float2(0.0f, 0.0f);
    {
      // window values come from LDS in batches of 8, one batch ahead of the multiply-adds that use it; the
      // scheduling barriers keep the compiler from hoisting every read to the top
      const float2 *rowp = L + (owner ? ph : 0) * RL + (R + R / 8) * grp;   // pad(R*grp + c) = 9*grp + c + (c >> 3)
      constexpr int NC = R + Q8 - 1, CB = 8, NB = (NC + CB - 1) / CB;
      float2 xb[2][CB];
#pragma unroll
      for (int i = 0; i < CB; i++) xb[0][i] = rowp[i + (i >> 3)];
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (b + 1 < NB) {
#pragma unroll
          for (int i = 0; i < CB; i++) {
            const int c = (b + 1) * CB + i;
            if (c < NC) xb[(b + 1) & 1][i] = rowp[c + (c >> 3)];
          }
        }
#pragma unroll
        for (int i = 0; i < CB; i++) {
          const int c = b * CB + i;
          if (c < NC) {
            const float2 x = xb[b & 1][i];
#pragma unroll
            for (int r = 0; r < R; r++) {
              const int q = r + Q8 - 1 - c;
              if (q >= 0 && q < Q8) {
                acc[r].x = __builtin_fmaf(gp[q], x.x, acc[r].x);
                acc[r].y = __builtin_fmaf(gp[q], x.y, acc[r].y);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- sum over the D component lanes (DPP), lane ph stores the outputs r = ph (mod D) -----------
#define CHAN_DPP_ADD(ctrl)                                                                                   \
    {                                                                                                        \
      _Pragma("unroll") for (int r = 0; r < R; r++) {                                                        \
        acc[r].x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[r].x), ctrl, 0xF, 0xF, false)); \
        acc[r].y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[r].y), ctrl, 0xF, 0xF, false)); \
      }                                                                                                      \
    }
    if (LG >= 2) CHAN_DPP_ADD(0xB1)     // quad_perm [1,0,3,2]
    if (LG >= 4) CHAN_DPP_ADD(0x4E)     // quad_perm [2,3,0,1]
    if (LG >= 8) CHAN_DPP_ADD(0x141)    // row_half_mirror
    if (LG >= 16) CHAN_DPP_ADD(0x140)   // row_mirror
#undef CHAN_DPP_ADD
    {
      // every lane of a group holds all R sums now; lane ph keeps the PER consecutive outputs ph*PER.. (a select
      // chain over compile-time register indices) so that a wave stores one contiguous run, 16 bytes per lane
      constexpr int PER = (LG <= R) ? R / LG : 1, ACT = (LG <= R) ? LG : R;   // outputs per lane, storing lanes per group
      float2 mine[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) {
        mine[i] = acc[i];
#pragma unroll
        for (int c = 1; c < ACT; c++) {
          if (ph == c) mine[i] = acc[c * PER + i];
        }
      }
      float2 *__restrict__ ob = p.out + t * TOB;                       // wave-uniform base, 32-bit lane offsets
      const unsigned long long left = p.n_out - t * TOB;
      const unsigned lim = left < (unsigned long long)TOB ? (unsigned)left : (unsigned)TOB;
      const unsigned o = (unsigned)(R * grp + ph * PER);
      if (ph < ACT) {
        if (o + PER <= lim) {
          if constexpr (PER == 1) ob[o] = mine[0];
          else {
#pragma unroll
            for (int i = 0; i < PER; i += 2)
              *reinterpret_cast<float4 *>(ob + o + i) = make_float4(mine[i].x, mine[i].y, mine[i + 1].x, mine[i + 1].y);
          }
        } else {
#pragma unroll
          for (int i = 0; i < PER; i++) if (o + i < lim) ob[o + i] = mine[i];
        }
      }
    }
    __syncthreads();                    // every lane has read its windows before the tile is overwritten
    t = t_next;
  }
}

// new history = last K-1 samples of (history ++ input)
__global__ void chan_carry_kernel(const float2 *in, long long n_in, const float2 *old_carry, float2 *new_carry, int k1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k1) return;
  const long long s = n_in - k1 + i;
  new_carry[i] = s >= 0 ? in[s] : old_carry[k1 + s];
}

}  // namespace lora_chan

using lora_chan::CParams;

struct lora_hip_chan {
  int device = 0;
  unsigned nfilts = 32, K = 0, dec_rate = 0;
  float rate = 0, flt_rate = 0;
  double phase_inc = 0, inc_eff = 0;
  // stream state (pfb_arb_resampler_ccf: d_last_filter, d_acc; + what a flat-array caller needs)
  unsigned j = 0, j_init = 0;
  float acc = 0;
  size_t skip = 0;                       // inputs of the next call already stepped over
  unsigned long long n_abs = 0;          // samples consumed so far (rotation phase)
  // d_flt_rate as an integer count of 2^-24 when the float recurrence is exact (see frac_pos below)
  bool frac = false;
  unsigned long long a = 0;
  bool force_general = false;
  hipStream_t own_stream = nullptr, stream = nullptr;
  float *d_filt = nullptr, *d_dfilt = nullptr;
  float2 *d_bank2 = nullptr;             // {f0, f1}[nfilts][K] for chan_arm_kernel
  bool no_arm = false;                   // developer knob LORA_CHAN_NO_ARM=1: keep chan_arb_kernel (A/B)
  unsigned arm_e = 0;                    // chan_arm_kernel: log2 of the tile's de-interleaving (arm_deinterleave)
  unsigned arm_tile = 0;                 // chan_arm_kernel: outputs per tile (arm_pick_tile); 0: the kernel is not used
  float arm_lane_use = 0;                // share of the filter loop's lanes that hold an output at that tile size
  float2 *d_carry[2] = {nullptr, nullptr}; int cur_carry = 0;
  float2 *d_rot_lane = nullptr;
  float2 *d_in = nullptr; size_t cap_in = 0;
  float2 *d_out = nullptr; size_t cap_out = 0;
  unsigned *d_spos = nullptr; float *d_sacc = nullptr; size_t cap_sched = 0;
  unsigned *h_spos = nullptr; float *h_sacc = nullptr; size_t cap_hsched = 0;   // pinned
  int num_cu = 256;
  char hip_err[256] = {0};
  char kernel_name[64] = {0};
};

namespace {

int hip_fail(lora_hip_chan *h, hipError_t e, const char *what) {
  if (h) snprintf(h->hip_err, sizeof h->hip_err, "%s: %s", what, hipGetErrorString(e));
  return LORA_HIP_EHIP;
}
#define CHAN_TRY(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail((h), e_, #call); } while (0)

struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess) { prev = -1; return; }
    if (prev == device) { prev = -1; return; }
    err = hipSetDevice(device);
    if (err != hipSuccess) prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

template <typename T>
int grow(lora_hip_chan *h, T **buf, size_t *cap, size_t want) {
  if (want <= *cap) return LORA_HIP_OK;
  if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
  const size_t ncap = want + want / 4;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(buf), ncap * sizeof(T));
  if (e != hipSuccess) { *buf = nullptr; hip_fail(h, e, "hipMalloc"); return LORA_HIP_ENOMEM; }
  *cap = ncap;
  return LORA_HIP_OK;
}

typedef void (*chan_fn)(const CParams);
struct DecimVariant { int D, nch; chan_fn fn[2]; int tob, ni; };
#define CHAN_DV(D, N) {D, N, {&lora_chan::chan_decim_kernel<D, N, false>, &lora_chan::chan_decim_kernel<D, N, true>}, \
                       lora_chan::DecimGeom<D, N>::TOB, lora_chan::DecimGeom<D, N>::NI}
const DecimVariant kDecim[] = {
    CHAN_DV(2, 2), CHAN_DV(2, 3), CHAN_DV(2, 4), CHAN_DV(4, 2), CHAN_DV(4, 3), CHAN_DV(4, 4),
    CHAN_DV(8, 2), CHAN_DV(8, 3), CHAN_DV(8, 4), CHAN_DV(16, 2), CHAN_DV(16, 3), CHAN_DV(16, 4),
    // decimations that are not a power of two (3 taps chunks = what the 100 dB design needs; others run chan_arb_kernel)
    CHAN_DV(3, 3), CHAN_DV(5, 3), CHAN_DV(6, 3), CHAN_DV(7, 3), CHAN_DV(10, 3), CHAN_DV(12, 3),
};
const DecimVariant *pick_decim(const lora_hip_chan *h) {
  if (h->force_general || h->flt_rate != 0.0f || h->dec_rate == 0 || h->dec_rate % h->nfilts) return nullptr;
  const unsigned D = h->dec_rate / h->nfilts;
  const unsigned q8 = (((h->K + D - 1) / D) + 7) / 8;
  for (const DecimVariant &v : kDecim) if ((unsigned)v.D == D && (unsigned)v.nch == (q8 < 2 ? 2 : q8)) return &v;
  return nullptr;
}

constexpr int kGeneralTileSamples = 6144;       // 48 KiB of LDS per workgroup
constexpr size_t kArbLdsBytes = 64 * 1024;      // chan_arb_kernel: banks + tile per workgroup
// the fast fractional-rate kernel needs both banks and a useful tile in LDS
inline size_t arb_bank_bytes(const lora_hip_chan *h) { return (size_t)h->nfilts * (h->K | 1u) * 8; }
inline bool arb_fits(const lora_hip_chan *h) {
  return h->frac && !h->force_general && arb_bank_bytes(h) + (size_t)(4 * h->K + 64) * 8 <= kArbLdsBytes;
}
// the arm-grouped kernel: closed-form positions, few enough arms for its LDS counters, and a tile that holds at least
// a quarter wave of outputs per arm beside their input span even when every arm occurs (below that the padding of the runs
// wastes more lanes than the wave-uniform taps gain)
constexpr size_t kArmLdsBytes = (160 * 1024) / lora_chan::kArmWgPerCu - (lora_chan::kArmBlock / 64) * 256 - 512;   // dynamic LDS per workgroup (its static counters come on top)
inline unsigned arm_span(const lora_hip_chan *h, unsigned t) {
  return (unsigned)(((unsigned long long)(t - 1) * (h->dec_rate + 1)) / h->nfilts + 2 + h->K);
}
chan_fn arm_kernel_fn(bool rot, bool interp, int e) {
#define ARM_E(E) (rot ? (interp ? &lora_chan::chan_arm_kernel<true, true, E> : &lora_chan::chan_arm_kernel<true, false, E>) \
                      : (interp ? &lora_chan::chan_arm_kernel<false, true, E> : &lora_chan::chan_arm_kernel<false, false, E>))
  return e == 0 ? ARM_E(0) : e == 1 ? ARM_E(1) : ARM_E(2);
#undef ARM_E
}
// largest t with 8*span(t) [tile] + 2*(t + 63*nfilts) [order] + 3*t [window start, arm] <= kArmLdsBytes (16-bit entries: t < 65535)
inline unsigned arm_tile_max(const lora_hip_chan *h) {
  unsigned lo = 0, hi = 60000;
  while (hi - lo > 1) {
    const unsigned mid = lo + (hi - lo) / 2;
    if (((size_t)arm_span(h, mid) + 8) * 8 + 2 * ((size_t)mid + 63 * h->nfilts) + 3 * (size_t)mid + 16 <= kArmLdsBytes) lo = mid; else hi = mid;
  }
  return lo;
}
inline unsigned arm_tile_out(const lora_hip_chan *h) { return h->arm_tile; }
inline bool arm_fits(const lora_hip_chan *h) {
  return h->frac && !h->force_general && !h->no_arm && h->nfilts <= lora_chan::kArmMaxFilts && h->arm_tile >= 64;
}
constexpr int kFracBits = lora_chan::kFracBits;
constexpr unsigned long long kFracMask = (1ull << kFracBits) - 1;

// The index recurrence of pfb_arb_resampler_ccf::filter, step by step (float d_acc, fmodf), over n_in inputs;
// stores (position, d_acc) per output when pos != nullptr.  Used where the closed form below does not hold.
struct Recur { unsigned j; float acc; size_t skip; };
size_t run_recurrence(const lora_hip_chan *h, Recur &st, size_t n_in, unsigned *pos, float *accs) {
  size_t i_out = 0, i_in = st.skip;
  unsigned j = st.j;
  float acc = st.acc;
  while (i_in < n_in) {
    while (j < h->nfilts) {
      if (pos) { pos[i_out] = (unsigned)(i_in * h->nfilts + j); accs[i_out] = acc; }
      i_out++;
      acc += h->flt_rate;
      j += h->dec_rate + (int)floor(acc);
      acc = fmodf(acc, 1.0f);
    }
    i_in += (size_t)(j / h->nfilts);
    j = j % h->nfilts;
  }
  st.j = j; st.acc = acc; st.skip = i_in - n_in;
  return i_out;
}

// Closed form of the same recurrence.  v = nfilts/rate is a float >= 1, so d_flt_rate = v - floor(v) is a
// multiple of ulp(v) >= 2^-23, d_acc + d_flt_rate < 2 has at most 24 significant bits, and every float
// operation of the recurrence (the add, floor, fmodf) is exact: d_acc is an integer count of 2^-24 that
// advances by a = d_flt_rate * 2^24 modulo 2^24, and the carries are what floor(d_acc) adds to j.  Output k
// after a state (P0 = nfilts*skip + j, A0) therefore sits at P0 + k*dec + ((A0 + k*a) >> 24) with weight
// ((A0 + k*a) mod 2^24) / 2^24: no sequential pass, and the kernel evaluates it per lane.
typedef unsigned __int128 u128;
inline u128 frac_pos(const lora_hip_chan *h, u128 p0, unsigned long long a0, u128 k) {
  return p0 + k * h->dec_rate + ((a0 + k * h->a) >> kFracBits);
}
// number of outputs over n_in inputs = first k with position >= nfilts*n_in (positions never decrease)
size_t frac_count(const lora_hip_chan *h, size_t n_in, unsigned long long *p0_out, unsigned long long *a0_out) {
  const u128 p0 = (u128)h->skip * h->nfilts + h->j;
  const unsigned long long a0 = (unsigned long long)ldexpf(h->acc, kFracBits);
  if (p0_out) *p0_out = (unsigned long long)p0;
  if (a0_out) *a0_out = a0;
  const u128 end = (u128)n_in * h->nfilts;
  if (end <= p0) return 0;
  u128 lo = 0, hi;                               // invariant: pos(lo) < end <= pos(hi)
  if (h->dec_rate) hi = (end - p0) / h->dec_rate + 1;
  else hi = (((end - p0) << kFracBits) / h->a) + 2;
  while (hi - lo > 1) {
    const u128 mid = lo + (hi - lo) / 2;
    if (frac_pos(h, p0, a0, mid) < end) lo = mid; else hi = mid;
  }
  return (size_t)hi;
}

// chan_arm_kernel's tile layout: the number S of inputs between consecutive outputs of one arm, over the stream's first
// outputs; when one S clearly dominates, the tile is de-interleaved by the power of two in it (at most 4 rows)
unsigned arm_deinterleave(const lora_hip_chan *h) {
  if (!h->frac || h->nfilts > lora_chan::kArmMaxFilts) return 0;
  std::vector<long long> last(h->nfilts, -1);
  std::vector<std::pair<unsigned long long, unsigned>> hist;       // (stride, count), a handful of entries
  unsigned pairs = 0;
  for (unsigned k = 0; k < 8192; k++) {
    const u128 P = frac_pos(h, h->j_init, 0, k);
    const unsigned arm = (unsigned)(P % h->nfilts);
    const long long in = (long long)(P / h->nfilts);
    if (last[arm] >= 0) {
      const unsigned long long sdiff = (unsigned long long)(in - last[arm]);
      pairs++;
      bool found = false;
      for (auto &e : hist) if (e.first == sdiff) { e.second++; found = true; break; }
      if (!found && hist.size() < 64) hist.push_back({sdiff, 1u});
    }
    last[arm] = in;
  }
  unsigned long long best = 0; unsigned best_n = 0;
  for (const auto &e : hist) if (e.second > best_n) { best = e.first; best_n = e.second; }
  if (best == 0 || best_n * 2 < pairs) return 0;
  const unsigned tz = (unsigned)__builtin_ctzll(best);
  return tz > 2 ? 2 : tz;
}

// chan_arm_kernel's tile size.  A tile's outputs are sorted into runs of one arm, each padded to whole waves, so the share of
// lanes that do useful work is t / sum_arm ceil64(count_arm): a rate that visits few arms (0.3: three or four) loses
// nothing, one that visits all of them evenly loses up to half unless a run is just under a multiple of 64 long.  The arm
// sequence is known in closed form, so the host tries the tile sizes that fit the LDS, counts the arms of the stream's first
// two tiles for each, and keeps the size with the most useful lanes per tile (large tiles win ties: fewer barriers per output;
// also counting whole rounds of the workgroup's eight waves was measured and changed nothing: the CU's other workgroup fills in).
void arm_pick_tile(lora_hip_chan *h) {
  h->arm_tile = 0; h->arm_lane_use = 0;
  if (!h->frac || h->nfilts > lora_chan::kArmMaxFilts) return;
  const unsigned tmax = arm_tile_max(h);
  if (tmax < 64) return;
  const unsigned n_sim = 2 * tmax;
  std::vector<unsigned char> arm(n_sim);
  for (unsigned k = 0; k < n_sim; k++) arm[k] = (unsigned char)(frac_pos(h, h->j_init, 0, k) % h->nfilts);
  std::vector<unsigned> cnt(h->nfilts);
  double best = 0; unsigned best_t = 0;
  for (unsigned t = tmax; t >= 64 && t >= tmax / 2; t -= (t > 16 ? 8 : t)) {
    unsigned lanes = 0;
    for (unsigned tile = 0; tile < 2; tile++) {
      std::fill(cnt.begin(), cnt.end(), 0u);
      for (unsigned k = tile * t; k < (tile + 1) * t; k++) cnt[arm[k]]++;
      for (unsigned c : cnt) lanes += (c + 63u) & ~63u;
    }
    const double use = 2.0 * t / (double)lanes;
    if (use > best * 1.02) { best = use; best_t = t; }          // a smaller tile must gain 2 % to be preferred
  }
  if (best < 0.45) return;                                        // below that chan_arb_kernel (one arm per lane) is the better kernel
  h->arm_tile = best_t; h->arm_lane_use = (float)best;
}

// fallback positions (i_in*nfilts + j, run_recurrence) are 32-bit: the inputs of one piece must keep them below 2^32
// for every nfilts create() accepts (a position is stored with j < nfilts, so it is below nfilts * piece <= 2^31)
inline size_t sched_chunk(unsigned nfilts) { return (size_t)((1ull << 31) / (nfilts ? nfilts : 1)); }

}  // namespace

extern "C" {

size_t lora_hip_chan_piece_limit(unsigned nfilts) { return sched_chunk(nfilts); }

int lora_hip_chan_design_taps(double rate, unsigned nfilts, double atten_db, float *taps, size_t cap, size_t *ntaps) {
  if (!ntaps || !(rate > 0.0) || nfilts == 0 || !(atten_db > 0.0) || (cap && !taps)) return LORA_HIP_EINVAL;
  if (rate >= 1.0) return LORA_HIP_EUNSUPPORTED;
  // pfb.py create_taps, rate < 1: the pass band is 80 % of the output's half band, the transition half of that
  const double half_band = 0.5 * rate, cutoff = 0.80 * half_band, transition = 0.40 * half_band;
  const double fs = (double)nfilts;
  // firdes::compute_ntaps_windes: atten*fs/(22*transition), made odd
  int n = (int)(atten_db * fs / (22.0 * transition));
  if (!(n & 1)) n++;
  if (n < 3) return LORA_HIP_EINVAL;
  *ntaps = (size_t)n;
  if (cap == 0) return LORA_HIP_OK;
  std::vector<float> t((size_t)n);
  // window.cc blackman_harris (92 dB set) through coswindow, evaluated on a float (ntaps - 1)
  const float span = (float)(n - 1);
  const int mid = (n - 1) / 2;
  const double w0 = 2.0 * M_PI * cutoff / fs;
  for (int i = 0; i < n; i++) {
    const float win = (float)(0.35875f - 0.48829f * std::cos((2.0 * M_PI * i) / span) +
                              0.14128f * std::cos((4.0 * M_PI * i) / span) - 0.01168f * std::cos((6.0 * M_PI * i) / span));
    const int k = i - mid;
    // firdes::low_pass_2: truncated sin(x)/x times the window
    t[(size_t)i] = (k == 0) ? (float)(w0 / M_PI * win) : (float)(std::sin(k * w0) / (k * M_PI) * win);
  }
  // unit gain at DC times the requested gain (= nfilts)
  double dc = t[(size_t)mid];
  for (int k = 1; k <= mid; k++) dc += 2 * t[(size_t)(mid + k)];
  const double scale = fs / dc;
  for (int i = 0; i < n; i++) t[(size_t)i] = (float)(t[(size_t)i] * scale);
  memcpy(taps, t.data(), sizeof(float) * (cap < (size_t)n ? cap : (size_t)n));
  return LORA_HIP_OK;
}

int lora_hip_chan_create(lora_hip_chan_t **out, int device, double phase_inc, float rate, const float *taps,
                         size_t ntaps, unsigned nfilts) {
  if (!out) return LORA_HIP_EINVAL;
  *out = nullptr;
  if (!taps || ntaps < 2 || nfilts == 0 || nfilts > 1024 || !(rate > 0.0f) || !std::isfinite(rate) || !std::isfinite(phase_inc))
    return LORA_HIP_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LORA_HIP_ENODEVICE;
  if (device < 0 || device >= ndev) return LORA_HIP_ENODEVICE;
  lora_hip_chan *h = new (std::nothrow) lora_hip_chan();
  if (!h) return LORA_HIP_ENOMEM;
  h->device = device; h->nfilts = nfilts; h->rate = rate; h->phase_inc = phase_inc;
  // pfb_arb_resampler_ccf ctor + set_rate: float arithmetic on an unsigned filter count
  const float v = nfilts / rate;
  if (!(v < 1.0e9f)) { delete h; return LORA_HIP_EINVAL; }
  h->dec_rate = (unsigned)floor(v);
  h->flt_rate = v - h->dec_rate;
  h->j_init = h->j = (unsigned)((ntaps / 2) % nfilts);
  h->K = (unsigned)ceil((double)ntaps / (double)nfilts);
  if (h->dec_rate == 0 && h->flt_rate == 0.0f) { delete h; return LORA_HIP_EINVAL; }
  {
    const float scaled = ldexpf(h->flt_rate, kFracBits);
    h->frac = (v >= 1.0f) && (scaled == floorf(scaled));
    h->a = h->frac ? (unsigned long long)scaled : 0;
  }
  // rotator: the per-sample factor is exp(i*(float)phase_inc) rounded to float and normalised; its angle is the
  // rotation the reference applies per sample
  if (phase_inc != 0.0) {
    const float pf = (float)phase_inc;
    float ir = cosf(pf), ii = sinf(pf);
    const float mag = hypotf(ir, ii);
    ir /= mag; ii /= mag;
    h->inc_eff = atan2((double)ii, (double)ir);
  }
  // create_taps / create_diff_taps / fir_filter::set_taps: bank[arm][K-1-q] = taps[arm + q*nfilts]
  const unsigned K = h->K;
  std::vector<float> bank((size_t)nfilts * K, 0.0f), dbank((size_t)nfilts * K, 0.0f);
  for (size_t i = 0; i < ntaps; i++) {
    const float d = (i + 1 < ntaps) ? (-1.0f * taps[i] + 1.0f * taps[i + 1]) : 0.0f;
    const size_t arm = i % nfilts, q = i / nfilts;
    bank[arm * K + (K - 1 - q)] = taps[i];
    dbank[arm * K + (K - 1 - q)] = d;
  }
  auto fail = [&](int code) { lora_hip_chan_destroy(h); return code; };
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(LORA_HIP_EHIP);
  h->stream = h->own_stream;
  const size_t bank_bytes = bank.size() * sizeof(float), carry_n = K > 1 ? K - 1 : 1;
  if (hipMalloc((void **)&h->d_filt, bank_bytes) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  if (hipMalloc((void **)&h->d_dfilt, bank_bytes) != hipSuccess) return fail(LORA_HIP_ENOMEM);
  for (int c = 0; c < 2; c++) {
    if (hipMalloc((void **)&h->d_carry[c], carry_n * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
    if (hipMemset(h->d_carry[c], 0, carry_n * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  if (hipMemcpy(h->d_filt, bank.data(), bank_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  if (hipMemcpy(h->d_dfilt, dbank.data(), bank_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  {
    std::vector<float2> b2(bank.size());
    for (size_t i = 0; i < bank.size(); i++) b2[i] = make_float2(bank[i], dbank[i]);
    if (hipMalloc((void **)&h->d_bank2, b2.size() * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
    if (hipMemcpy(h->d_bank2, b2.data(), b2.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  if (const char *e = getenv("LORA_CHAN_NO_ARM")) h->no_arm = atoi(e) != 0;
  h->arm_e = arm_deinterleave(h);
  arm_pick_tile(h);
  if (const char *e = getenv("LORA_CHAN_ARM_E")) { const int v2 = atoi(e); if (v2 >= 0 && v2 <= 2) h->arm_e = (unsigned)v2; }   // developer knob (A/B)
  {
    std::vector<float2> lane(lora_chan::kBlock);
    for (int t = 0; t < lora_chan::kBlock; t++) lane[t] = make_float2((float)std::cos(h->inc_eff * t), (float)std::sin(h->inc_eff * t));
    if (hipMalloc((void **)&h->d_rot_lane, lane.size() * sizeof(float2)) != hipSuccess) return fail(LORA_HIP_ENOMEM);
    if (hipMemcpy(h->d_rot_lane, lane.data(), lane.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(LORA_HIP_EHIP);
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (K > (unsigned)kGeneralTileSamples - 64) return fail(LORA_HIP_EUNSUPPORTED);
  {
    const void *fns[] = {(const void *)&lora_chan::chan_general_kernel<false, 0>, (const void *)&lora_chan::chan_general_kernel<true, 0>,
                         (const void *)&lora_chan::chan_general_kernel<false, 1>, (const void *)&lora_chan::chan_general_kernel<true, 1>,
                         (const void *)&lora_chan::chan_general_kernel<false, 2>, (const void *)&lora_chan::chan_general_kernel<true, 2>};
    for (const void *f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kGeneralTileSamples * 8) != hipSuccess) return fail(LORA_HIP_EHIP);
    const void *arb[] = {(const void *)&lora_chan::chan_arb_kernel<false, false>, (const void *)&lora_chan::chan_arb_kernel<true, false>,
                         (const void *)&lora_chan::chan_arb_kernel<false, true>, (const void *)&lora_chan::chan_arb_kernel<true, true>};
    for (const void *f : arb)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kArbLdsBytes) != hipSuccess) return fail(LORA_HIP_EHIP);
    for (int i = 0; i < 12; i++)
      if (hipFuncSetAttribute((const void *)arm_kernel_fn(i & 1, (i >> 1) & 1, i >> 2), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kArmLdsBytes) != hipSuccess) return fail(LORA_HIP_EHIP);
  }
  lora_hip_chan_force_general(h, 0);
  *out = h;
  return LORA_HIP_OK;
}

int lora_hip_chan_force_general(lora_hip_chan_t *h, int on) {
  if (!h) return LORA_HIP_EINVAL;
  h->force_general = on != 0;
  const DecimVariant *v = pick_decim(h);
  if (v) snprintf(h->kernel_name, sizeof h->kernel_name, "chan_decim_kernel<%d,%d>", v->D, v->nch);
  else if (arm_fits(h)) snprintf(h->kernel_name, sizeof h->kernel_name, "chan_arm_kernel");
  else if (arb_fits(h)) snprintf(h->kernel_name, sizeof h->kernel_name, "chan_arb_kernel");
  else snprintf(h->kernel_name, sizeof h->kernel_name, "chan_general_kernel");
  return LORA_HIP_OK;
}

void lora_hip_chan_destroy(lora_hip_chan_t *h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); (void)hipStreamDestroy(h->own_stream); }
  (void)hipFree(h->d_filt); (void)hipFree(h->d_dfilt); (void)hipFree(h->d_bank2); (void)hipFree(h->d_carry[0]); (void)hipFree(h->d_carry[1]);
  (void)hipFree(h->d_rot_lane); (void)hipFree(h->d_in); (void)hipFree(h->d_out); (void)hipFree(h->d_spos); (void)hipFree(h->d_sacc);
  if (h->h_spos) (void)hipHostFree(h->h_spos);
  if (h->h_sacc) (void)hipHostFree(h->h_sacc);
  delete h;
}

int lora_hip_chan_set_stream(lora_hip_chan_t *h, void *hip_stream) {
  if (!h) return LORA_HIP_EINVAL;
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return LORA_HIP_OK;
}

const char *lora_hip_chan_last_hip_error(const lora_hip_chan_t *h) { return h ? h->hip_err : ""; }
unsigned lora_hip_chan_taps_per_filter(const lora_hip_chan_t *h) { return h ? h->K : 0; }
unsigned lora_hip_chan_dec_rate(const lora_hip_chan_t *h) { return h ? h->dec_rate : 0; }
float lora_hip_chan_flt_rate(const lora_hip_chan_t *h) { return h ? h->flt_rate : 0.0f; }
const char *lora_hip_chan_kernel_name(const lora_hip_chan_t *h) { return h ? h->kernel_name : ""; }
double lora_hip_chan_phase_inc_effective(const lora_hip_chan_t *h) { return h ? h->inc_eff : 0.0; }

int lora_hip_chan_reset(lora_hip_chan_t *h) {
  if (!h) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  CHAN_TRY(h, guard.err);
  h->j = h->j_init; h->acc = 0; h->skip = 0; h->n_abs = 0;
  const size_t carry_n = h->K > 1 ? h->K - 1 : 1;
  CHAN_TRY(h, hipMemsetAsync(h->d_carry[h->cur_carry], 0, carry_n * sizeof(float2), h->stream));
  CHAN_TRY(h, hipStreamSynchronize(h->stream));
  return LORA_HIP_OK;
}

int lora_hip_chan_seek(lora_hip_chan_t *h, uint64_t n_consumed, const lora_c32 *history, unsigned flags) {
  if (!h || (flags & ~(unsigned)LORA_HIP_IQ_ON_DEVICE) || n_consumed > ((uint64_t)1 << 56)) return LORA_HIP_EINVAL;
  DeviceGuard guard(h->device);
  CHAN_TRY(h, guard.err);
  // the state after n_consumed samples of a fresh stream
  h->j = h->j_init; h->acc = 0; h->skip = 0;
  if (n_consumed) {
    if (h->frac) {
      unsigned long long p0 = 0, a0 = 0;
      const size_t cnt = frac_count(h, (size_t)n_consumed, &p0, &a0);
      const u128 pn = frac_pos(h, p0, a0, cnt) - (u128)n_consumed * h->nfilts;
      h->j = (unsigned)(pn % h->nfilts); h->skip = (size_t)(pn / h->nfilts);
      h->acc = ldexpf((float)((a0 + (unsigned long long)cnt * h->a) & kFracMask), -kFracBits);
    } else {
      Recur st{h->j, h->acc, h->skip};
      run_recurrence(h, st, (size_t)n_consumed, nullptr, nullptr);
      h->j = st.j; h->acc = st.acc; h->skip = st.skip;
    }
  }
  h->n_abs = n_consumed;
  const size_t k1 = h->K > 1 ? h->K - 1 : 0;
  float2 *carry = h->d_carry[h->cur_carry];
  if (k1) {
    if (!history) CHAN_TRY(h, hipMemsetAsync(carry, 0, k1 * sizeof(float2), h->stream));
    else CHAN_TRY(h, hipMemcpyAsync(carry, history, k1 * sizeof(float2),
                                    (flags & LORA_HIP_IQ_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    CHAN_TRY(h, hipStreamSynchronize(h->stream));
  }
  return LORA_HIP_OK;
}

size_t lora_hip_chan_output_count(lora_hip_chan_t *h, size_t n_in) {
  if (!h) return 0;
  if (h->frac) return frac_count(h, n_in, nullptr, nullptr);
  Recur st{h->j, h->acc, h->skip};
  return run_recurrence(h, st, n_in, nullptr, nullptr);
}

static int chan_work_chunk(lora_hip_chan_t *h, const float2 *d_in, size_t n_in, float2 *d_out, size_t *n_out) {
  const bool rot = h->phase_inc != 0.0;
  CParams p;
  memset(&p, 0, sizeof p);
  p.in = d_in; p.carry = h->d_carry[h->cur_carry]; p.n_in = (long long)n_in; p.out = d_out;
  p.filt = h->d_filt; p.dfilt = h->d_dfilt; p.K = h->K; p.nfilts = h->nfilts; p.dec = h->dec_rate; p.a = h->a;
  p.inc = h->inc_eff;
  p.phase0 = rot ? (double)fmodl((long double)h->inc_eff * (long double)h->n_abs, 2.0L * 3.14159265358979323846264338327950288L) : 0.0;
  size_t cnt;
  Recur st{h->j, h->acc, h->skip};
  int mode;
  if (h->frac) {
    cnt = frac_count(h, n_in, &p.p0, &p.a0);
    const u128 pn = frac_pos(h, p.p0, p.a0, cnt) - (u128)n_in * h->nfilts;      // position of the next output
    st.j = (unsigned)(pn % h->nfilts); st.skip = (size_t)(pn / h->nfilts);
    st.acc = ldexpf((float)((p.a0 + (unsigned long long)cnt * h->a) & kFracMask), -kFracBits);
    mode = (h->a == 0 && p.a0 == 0) ? 0 : 1;
  } else {
    mode = 2;
    cnt = run_recurrence(h, st, n_in, nullptr, nullptr);
    if (cnt) {
      // the pinned arrays may still feed the previous call's copy
      CHAN_TRY(h, hipStreamSynchronize(h->stream));
      if (cnt > h->cap_hsched) {
        if (h->h_spos) (void)hipHostFree(h->h_spos);
        if (h->h_sacc) (void)hipHostFree(h->h_sacc);
        h->h_spos = nullptr; h->h_sacc = nullptr; h->cap_hsched = 0;
        const size_t ncap = cnt + cnt / 4;
        if (hipHostMalloc((void **)&h->h_spos, ncap * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) return LORA_HIP_ENOMEM;
        if (hipHostMalloc((void **)&h->h_sacc, ncap * sizeof(float), hipHostMallocDefault) != hipSuccess) return LORA_HIP_ENOMEM;
        h->cap_hsched = ncap;
      }
      if (cnt > h->cap_sched) {
        size_t c1 = h->cap_sched, c2 = h->cap_sched;
        int rc;
        if ((rc = grow(h, &h->d_spos, &c1, cnt)) != LORA_HIP_OK) { h->cap_sched = 0; return rc; }
        if ((rc = grow(h, &h->d_sacc, &c2, cnt)) != LORA_HIP_OK) { h->cap_sched = 0; return rc; }
        h->cap_sched = c1 < c2 ? c1 : c2;
      }
      Recur st2{h->j, h->acc, h->skip};
      run_recurrence(h, st2, n_in, h->h_spos, h->h_sacc);
      CHAN_TRY(h, hipMemcpyAsync(h->d_spos, h->h_spos, cnt * sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
      CHAN_TRY(h, hipMemcpyAsync(h->d_sacc, h->h_sacc, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
      p.sched_pos = h->d_spos; p.sched_acc = h->d_sacc;
    }
  }
  p.n_out = cnt;
  if (cnt) {
    const DecimVariant *v = (mode == 0) ? pick_decim(h) : nullptr;
    if (v) {
      p.i0 = (long long)(p.p0 / h->nfilts); p.jfix = (unsigned)(p.p0 % h->nfilts);
      const unsigned long long tiles = (cnt + v->tob - 1) / v->tob;
      unsigned blocks = (unsigned)h->num_cu * 4;
      if (tiles < blocks) blocks = (unsigned)tiles;
      if (rot) {
        const long double a = (long double)h->inc_eff * (long double)v->D * (long double)v->tob * (long double)blocks;
        p.step_re = (double)cosl(a); p.step_im = (double)sinl(a);
        for (int k = 0; k < lora_chan::kMaxSteps; k++) {
          const double b = h->inc_eff * 256.0 * k;
          p.rot_step[k] = make_float2((float)std::cos(b), (float)std::sin(b));
        }
        p.rot_lane = h->d_rot_lane;
      }
      hipLaunchKernelGGL(v->fn[rot ? 1 : 0], dim3(blocks), dim3(lora_chan::kBlock), 0, h->stream, p);
    } else if (mode != 2 && arm_fits(h)) {
      // outputs grouped by arm, taps through the scalar path: the whole dynamic LDS is tile + order list
      const unsigned to = arm_tile_out(h);
      const unsigned e = h->arm_e, span = arm_span(h, to);
      p.tile_out = to; p.bank2 = h->d_bank2;
      p.row_pitch = ((span + (1u << e) - 1) >> e) + 1;             // one spare slot per row: a window may end one row-slot further
      p.tile_cap = p.row_pitch << e;
      const unsigned long long tiles = (cnt + to - 1) / to;
      unsigned blocks = (unsigned)h->num_cu * lora_chan::kArmWgPerCu;
      if (tiles < blocks) blocks = (unsigned)tiles;
      hipLaunchKernelGGL(arm_kernel_fn(rot, mode == 1, (int)e), dim3(blocks), dim3(lora_chan::kArmBlock), kArmLdsBytes, h->stream, p);
    } else if (mode != 2 && arb_fits(h)) {
      // banks in LDS: outputs per tile = as many as keep the tile's input span in what is left
      const size_t tile_samples = (kArbLdsBytes - arb_bank_bytes(h)) / 8;
      unsigned to = lora_chan::kBlock;
      auto span_of = [&](unsigned t) { return ((unsigned long long)(t - 1) * (h->dec_rate + 1)) / h->nfilts + 2 + h->K; };
      while (to > 1 && span_of(to) > (unsigned long long)tile_samples) to /= 2;
      if (span_of(to) > (unsigned long long)tile_samples) return LORA_HIP_EUNSUPPORTED;   // arb_fits leaves room for 4K samples
      p.tile_out = to;
      const unsigned long long tiles = (cnt + to - 1) / to;
      unsigned blocks = (unsigned)h->num_cu * 2;
      if (tiles < blocks) blocks = (unsigned)tiles;
      chan_fn fn;
      if (mode == 1) fn = rot ? &lora_chan::chan_arb_kernel<true, true> : &lora_chan::chan_arb_kernel<false, true>;
      else fn = rot ? &lora_chan::chan_arb_kernel<true, false> : &lora_chan::chan_arb_kernel<false, false>;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(lora_chan::kBlock), kArbLdsBytes, h->stream, p);
    } else {
      // outputs per tile: as many as keep the tile's input span inside the LDS tile
      unsigned to = lora_chan::kBlock;
      auto span_of = [&](unsigned t) { return ((unsigned long long)(t - 1) * (h->dec_rate + 1)) / h->nfilts + 2 + h->K; };
      while (to > 1 && span_of(to) > (unsigned long long)kGeneralTileSamples) to /= 2;
      if (span_of(to) > (unsigned long long)kGeneralTileSamples) return LORA_HIP_EUNSUPPORTED;
      p.tile_out = to;
      const unsigned long long tiles = (cnt + to - 1) / to;
      unsigned blocks = (unsigned)h->num_cu * 8;
      if (tiles < blocks) blocks = (unsigned)tiles;
      const size_t lds = (size_t)kGeneralTileSamples * 8;
      chan_fn fn;
      if (mode == 0) fn = rot ? &lora_chan::chan_general_kernel<true, 0> : &lora_chan::chan_general_kernel<false, 0>;
      else if (mode == 1) fn = rot ? &lora_chan::chan_general_kernel<true, 1> : &lora_chan::chan_general_kernel<false, 1>;
      else fn = rot ? &lora_chan::chan_general_kernel<true, 2> : &lora_chan::chan_general_kernel<false, 2>;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(lora_chan::kBlock), lds, h->stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
  }
  if (n_in && h->K > 1) {
    const int k1 = (int)h->K - 1;
    hipLaunchKernelGGL(lora_chan::chan_carry_kernel, dim3((k1 + 255) / 256), dim3(256), 0, h->stream, d_in, (long long)n_in,
                       h->d_carry[h->cur_carry], h->d_carry[h->cur_carry ^ 1], k1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
    h->cur_carry ^= 1;
  }
  h->j = st.j; h->acc = st.acc; h->skip = st.skip; h->n_abs += n_in;
  *n_out = cnt;
  return LORA_HIP_OK;
}

int lora_hip_chan_work(lora_hip_chan_t *h, const lora_c32 *in, size_t n_in, lora_c32 *out, size_t out_cap,
                       size_t *n_out, unsigned flags) {
  if (!h || !n_out || (!in && n_in)) return LORA_HIP_EINVAL;
  *n_out = 0;
  const bool in_dev = flags & LORA_HIP_IQ_ON_DEVICE, out_dev = flags & LORA_HIP_OUT_ON_DEVICE;
  if ((flags & LORA_HIP_ASYNC) && (!in_dev || !out_dev)) return LORA_HIP_EINVAL;
  if (flags & ~(unsigned)(LORA_HIP_IQ_ON_DEVICE | LORA_HIP_OUT_ON_DEVICE | LORA_HIP_ASYNC)) return LORA_HIP_EINVAL;
  if (n_in > ((size_t)1 << 56)) return LORA_HIP_EINVAL;
  const size_t total = lora_hip_chan_output_count(h, n_in);
  if (total > out_cap || (total && !out)) return LORA_HIP_EINVAL;
  if (n_in == 0) return LORA_HIP_OK;
  DeviceGuard guard(h->device);
  CHAN_TRY(h, guard.err);
  int rc;
  const float2 *d_in = reinterpret_cast<const float2 *>(in);
  if (!in_dev) {
    if ((rc = grow(h, &h->d_in, &h->cap_in, n_in)) != LORA_HIP_OK) return rc;
    CHAN_TRY(h, hipMemcpyAsync(h->d_in, in, n_in * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    d_in = h->d_in;
  }
  float2 *d_out = reinterpret_cast<float2 *>(out);
  if (!out_dev) {
    if ((rc = grow(h, &h->d_out, &h->cap_out, total ? total : 1)) != LORA_HIP_OK) return rc;
    d_out = h->d_out;
  }
  // the fallback form carries 32-bit positions: long inputs go through in pieces (any chunking gives the same stream)
  const size_t piece = h->frac ? n_in : sched_chunk(h->nfilts);
  size_t done_in = 0, done_out = 0;
  while (done_in < n_in) {
    const size_t n = (n_in - done_in < piece) ? n_in - done_in : piece;
    size_t got = 0;
    if ((rc = chan_work_chunk(h, d_in + done_in, n, d_out + done_out, &got)) != LORA_HIP_OK) return rc;
    done_in += n; done_out += got;
  }
  if (!out_dev) {
    if (done_out) CHAN_TRY(h, hipMemcpyAsync(out, d_out, done_out * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
    CHAN_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!(flags & LORA_HIP_ASYNC)) {
    CHAN_TRY(h, hipStreamSynchronize(h->stream));
  }
  *n_out = done_out;
  return LORA_HIP_OK;
}

}  // extern "C"
