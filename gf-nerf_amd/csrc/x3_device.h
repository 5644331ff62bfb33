// x3_device.h -- device-side pieces of the split-f16 ("x3") decoder shared by field_x3.hip and fused_fwd.hip:
// operand split, the three-MFMA product, accumulator-tile -> operand-fragment conversion, plane stores, and the
// 32-sample decoder tile itself.  See field_x3.hip for the scheme.
#pragma once
#include "gfn_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int FWD_CHUNKS = 72, BWD_CHUNKS = 64, FWD_BIAS = 448;
static_assert(GFN_FIELD_X3_FWD_BYTES == FWD_CHUNKS * 2048 + FWD_BIAS * 4, "forward image size");
static_assert(GFN_FIELD_X3_BWD_BYTES == BWD_CHUNKS * 2048, "backward image size");
static_assert(GFN_FIELD_X3_FWD_BYTES <= 160 * 1024, "forward LDS image must fit one CU");

// forward chunk bases
constexpr int C_B0 = 0;    // base L1: [u 2][ot 4]
constexpr int C_B1 = 8;    // base L2: [u 8]
constexpr int C_H0 = 16;   // head L1: [u 4][ot 4]   u0 = geo rows of the base tile, u1 = SH, u2,u3 = appearance
constexpr int C_H1 = 32;   // head L2: [u 8][ot 4]
constexpr int C_H2 = 64;   // head L3: [u 8]
// backward chunk bases (A = W^T: rows = layer inputs, k = layer outputs)
constexpr int D_G3 = 0;    // d c2 = hW2^T dz3 : [it 4]
constexpr int D_G2 = 4;    // d c1 = hW1^T dz2 : [u 8][it 4]
constexpr int D_G1 = 36;   // d h[1:16] = hW0[:,16:31]^T dz1 : [u 8]
constexpr int D_GB1 = 44;  // d a1 = bW1^T dh : [it 4]
constexpr int D_GB0 = 48;  // d feat = bW0^T dz : [u 8]
constexpr int D_GE = 56;   // d appearance = hW0[:,31:63]^T dz1 : [u 8]   (global stage only)
// split-f16 planes kept for the weight gradients (global stage): one hi and one lo' buffer of cap * LD f16 each,
// PLANE-MAJOR: the plane with first column C and width W is a contiguous array of cap * W elements at element offset cap * C,
// and inside it TILE-MAJOR: [cap / 32 tiles][W / 8 octets][32 samples][8 features] (plane_off).  A wave of the decoder kernels
// holds 32 samples on its lanes and 4 (or 8) consecutive features of each in one register pair, so one store instruction
// writes 32 samples x 16 bytes = 512 contiguous bytes (sample-major rows made it 32 pieces of 16 bytes 2 * W bytes apart:
// 1.9 GB per step written as 120 M partial-line requests); gfn_wgrad_x3 reads the same 16-byte pieces back.  Columns:
constexpr int FP_LD = GFN_FIELD_X3_FWD_PLANE_LD;   // feat 0..31 | a1 32..159 | xp 160..223 | c1 224..351 | c2 352..479
constexpr int FP_FEAT = 0, FP_A1 = 32, FP_XP = 160, FP_C1 = 224, FP_C2 = 352;
constexpr int BP_LD = GFN_FIELD_X3_BWD_PLANE_LD;   // dz3 0..7 | dh 8..23 | dz2 24..151 | dz1 152..279 | dza1 280..407
constexpr int BP_DZ3 = 0, BP_DH = 8, BP_DZ2 = 24, BP_DZ1 = 152, BP_DZA1 = 280;
constexpr int BP_DZA1_LO = GFN_FIELD_X3_BWD_PLANE_LD, BP_LD_RNE = GFN_FIELD_X3_BWD_PLANE_LD_RNE;  // one-plane form: dza1's lo' plane behind the five
static_assert(FP_LD == 480 && BP_LD == 408 && BP_LD_RNE == BP_LD + 128, "plane row widths");
// PLANES (template parameter of the producers): 0 none; 1 both halves (hi = rtz, lo' = the residual: wgrad at x3 accuracy);
// 2 ONE plane per operand: x rounded to f16 once, to nearest even (v_cvt_pk_f16_f32), written where mode 1 writes hi.  The weight
// gradient is a sum over every sample of the batch (dW = sum_s dz a, 5e5 terms): a rounding that is zero-mean per term averages
// out in it (rtz would not: it shrinks every operand by 2^-12 on average), unlike in the per-sample chains d x = W^T dz, which stay
// x3.  Half the plane bytes written, half read by gfn_wgrad_x3_layers, one MFMA per fragment pair instead of three.
// One layer is kept out of it: base layer 1, dW = sum dza1 feat.  Its activations are the hash features -- f16 numbers already (one
// table in the global stage: the plane holds them exactly) and on a young table signed noise, so that dW is a sum of random-sign
// terms that is no larger than its own rounding noise times 2^12 (measured: 1.0e-4 of the largest |dW| from dz's rounding alone,
// whatever the dither).  dza1 therefore keeps its lo' plane (column BP_DZA1_LO of the same buffer, 128 wide) and neither it nor
// feat is dithered: that layer's gradient is the x3 one.
// Dither (PL_RNE): rounding alone is zero-mean but not independent from sample to sample -- neighbouring samples of a ray have
// nearly the same activations (the SH and appearance inputs are the SAME along a ray) and so nearly the same rounding error, which
// then adds up along the ray instead of averaging out (measured on the bench's global step: the activations' rounding contributed
// 3x what dz's did, 2e-4 of the largest |dW| in the head's first layer).  So sample q's activations are stored as
// rne(a * (1 + u_q)) and its dz as rne(dz * (1 - u_q)), u_q a fixed low-discrepancy hash of q in +-2^-11 (about one f16 ulp): the
// products dz a keep their value to 2^-22, every rounding meets its own position on the f16 grid, and db = sum dz (1 - u_q) gains
// zero-mean noise of 3e-4 per term.  amp = 0 switches it off (gfn_launch_opts.plain_rounding).
constexpr int PL_NONE = 0, PL_SPLIT = 1, PL_RNE = 2;
__device__ __forceinline__ float plane_dither(int64_t q, float amp) {
  const uint32_t hq = (uint32_t)q * 2654435761u;  // golden-ratio sequence: consecutive samples spread evenly
  return ((float)((int)(hq >> 20) - 2048) + 0.5f) * amp;  // amp = 2^-22: u in (-2^-11, 2^-11)
}

__device__ __forceinline__ int acc_row(int t, int s, int h, int i) { return 32 * t + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3); }

// element offset of features 8 * octet .. 8 * octet + 7 of sample q inside a tile-major plane of width W
__host__ __device__ __forceinline__ int64_t plane_off(int64_t q, int W, int octet) { return (((q >> 5) * (W >> 3) + octet) * 32 + (q & 31)) * 8; }

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const auto hv = __builtin_amdgcn_cvt_pkrtz(x0, x1);  // round toward zero: saturates at 65504, never rounds to inf
  // (x - hi) * 2048, exact: x - hi has at most 13 significant bits and 2048 is a power of two.  Written as fma(hi, -2048, x * 2048) -- the same
  // exact value, rounded (not at all) once -- because that form is ONE v_fma_mix_f32 per element (the f16 operand is read in place) behind one
  // packed multiply per pair, where (x - float(hi)) * 2048 costs a conversion, a subtraction and a multiplication per element.
  using f2 = __attribute__((ext_vector_type(2))) float;
  const f2 t = f2{x0, x1} * 2048.f;
  const float r0 = __builtin_fmaf((float)hv[0], -2048.f, t[0]), r1 = __builtin_fmaf((float)hv[1], -2048.f, t[1]);
  hi = __builtin_bit_cast(uint32_t, hv);
  lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// two f32 -> packed f16, round to nearest even: one v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t rne2(float x0, float x1) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x0, x1}, h2));
}
__device__ __forceinline__ void rne8(const float* x, u32x4& r, float pc = 1.f) {  // pc: the sample's dither factor (plane_dither)
#pragma unroll
  for (int p = 0; p < 4; p++) r[p] = rne2(x[2 * p] * pc, x[2 * p + 1] * pc);
}

struct Acc2 {
  f32x16 m, c;
};

__device__ __forceinline__ void mma3(Acc2& a, const u32x4* __restrict__ Ihi, const u32x4* __restrict__ Ilo, int chunk, int lane,
                                     const u32x4& bhi, const u32x4& blo) {
  const h8 ah = __builtin_bit_cast(h8, Ihi[chunk * 64 + lane]);
  const h8 al = __builtin_bit_cast(h8, Ilo[chunk * 64 + lane]);
  const h8 bh = __builtin_bit_cast(h8, bhi), bl = __builtin_bit_cast(h8, blo);
  a.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a.m, 0, 0, 0);
  a.c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a.c, 0, 0, 0);
  a.c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a.c, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting every later operand read (register blow-up)
}

// The same with the weight fragments of the NEXT chunk requested before this chunk's MFMAs are issued: read one chunk ahead, the LDS round trip
// (ds_read_b128 x 2, ~100 cycles; issued behind the second MFMA, where the current `hi` fragment's registers are free again) runs under the
// matrix pipe's work instead of in front of every triple --
// mma3 alone leaves the pipe idle for most of a round trip per chunk, because the barrier that keeps the operand reads from all being hoisted
// (register blow-up) also keeps the next read behind the MFMAs.  `w` carries the fragments from call to call; chunk_next < 0: nothing to prefetch.
struct WFrag {
  h8 hi, lo;
};
__device__ __forceinline__ WFrag wfrag_load(const u32x4* __restrict__ Ihi, const u32x4* __restrict__ Ilo, int chunk, int lane) {
  WFrag w;
  w.hi = __builtin_bit_cast(h8, Ihi[chunk * 64 + lane]);
  w.lo = __builtin_bit_cast(h8, Ilo[chunk * 64 + lane]);
  return w;
}
// X2 (a measurement, GFN_RENDER_X2: the activation operand as ONE f16 number, rounded to nearest -- weights stay split): two MFMAs per product sum
template <bool X2 = false>
__device__ __forceinline__ void mma3p(Acc2& a, WFrag& w, const u32x4* __restrict__ Ihi, const u32x4* __restrict__ Ilo, int chunk_next, int lane,
                                      const u32x4& bhi, const u32x4& blo) {
  const h8 bh = __builtin_bit_cast(h8, bhi), bl = __builtin_bit_cast(h8, blo);
  a.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, bh, a.m, 0, 0, 0);
  if (!X2) a.c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, bl, a.c, 0, 0, 0);
  h8 nh = w.hi, nl = w.lo;
  if (chunk_next >= 0) { nh = __builtin_bit_cast(h8, Ihi[chunk_next * 64 + lane]); }
  a.c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.lo, bh, a.c, 0, 0, 0);
  if (chunk_next >= 0) { nl = __builtin_bit_cast(h8, Ilo[chunk_next * 64 + lane]); }
  __builtin_amdgcn_sched_barrier(0);
  w.hi = nh; w.lo = nl;
}

// bias_h = the layer biases' LDS block + 4 * h (ONE lane-dependent address for all five layers: every load below is it + a compile-time offset;
// a pointer per layer cost the register-tight inference kernels five address registers), off = the layer's offset in floats
__device__ __forceinline__ void acc_init_bias(Acc2& a, const float* __restrict__ bias_h, int off, int ot) {
#pragma unroll
  for (int g = 0; g < 4; g++) {  // registers 4g..4g+3 = rows 32ot + 8g + 4h + 0..3
    const float4 b = *reinterpret_cast<const float4*>(bias_h + (off + 32 * ot + 8 * g));
    a.m[4 * g] = b.x; a.m[4 * g + 1] = b.y; a.m[4 * g + 2] = b.z; a.m[4 * g + 3] = b.w;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) a.c[r] = 0.f;
}
__device__ __forceinline__ void acc_init_zero(Acc2& a) {
#pragma unroll
  for (int r = 0; r < 16; r++) { a.m[r] = 0.f; a.c[r] = 0.f; }
}
__device__ __forceinline__ float acc_get(const Acc2& a, int r) { return __builtin_fmaf(a.c[r], 0x1p-11f, a.m[r]); }
// the same for all 16 registers of a tile: register pairs (2p, 2p + 1) through one v_pk_fma_f32 each (the same fma per element)
__device__ __forceinline__ void acc_get16(const Acc2& a, float (&v)[16]) {
  using f2 = __attribute__((ext_vector_type(2))) float;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const f2 r = __builtin_elementwise_fma(f2{a.c[2 * p], a.c[2 * p + 1]}, f2{0x1p-11f, 0x1p-11f}, f2{a.m[2 * p], a.m[2 * p + 1]});
    v[2 * p] = r[0]; v[2 * p + 1] = r[1];
  }
}

// a finished tile -> (ReLU, mask bits,) the two k-step fragments of the next layer
struct Frag {
  u32x4 hi[2], lo[2];
};
// where a tile's 32 columns go in a one-plane (PL_RNE) buffer: columns col0 .. col0+31 of sample q in the plane `base` of width W
// (registers 8s+4g..+3 = rows 16s+8g+4h..+3: lane half h owns 4 of the 8 features of octet (col0 + 16s + 8g) / 8); pc = the
// sample's dither factor.  Stored from inside the conversion, value by value: a tile's rounded copy never occupies registers
struct PlaneDst {
  uint16_t* base; int W; int64_t q; int col0; int h; float pc; bool on;
};
__device__ __forceinline__ void store_tile_rne(const PlaneDst& d, const float (&v)[16]) {
  if (!d.on) return;
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const float* x = v + 8 * s + 4 * g;
      *reinterpret_cast<uint2*>(d.base + plane_off(d.q, d.W, (d.col0 >> 3) + 2 * s + g) + 4 * d.h) =
          make_uint2(rne2(x[0] * d.pc, x[1] * d.pc), rne2(x[2] * d.pc, x[3] * d.pc));
    }
}
// RNE: the tile's 16 values also go, rounded to nearest, to the one-plane buffer row `dst` names
template <bool RELU, bool RNE = false, bool X2 = false>
__device__ __forceinline__ void tile_to_frag(const Acc2& a, Frag& f, unsigned long long& mask, int t, const PlaneDst* dst = nullptr) {
  float v[16];
  acc_get16(a, v);
  uint32_t bits = 0;  // this tile's 16 ReLU bits.  A C++ compare leaves a 64-bit lane mask in an SGPR pair, and with 16 of them per
                      // tile alive at once the compiler spilled SGPRs through v_writelane / v_readlane (724 of the forward kernel's
                      // ~2,700 vector instructions per tile).  Here every compare lands in VCC and is shifted into `bits` by the
                      // next instruction (bits = 2 * bits + carry), registers 15 .. 0 so that bit r ends up as register r's.
  // X2 without planes (inference): the ReLU is taken on the rounded pair instead -- max(rne(v), 0) = rne(max(v, 0)), rounding is monotone and keeps the sign --
  // one v_pk_max_f16 per pair in place of two v_max_f32 (NaN -> 0 either way)
  constexpr bool RELU_PACKED = RELU && X2 && !RNE;
#pragma unroll
  for (int r = 15; r >= 0; r--) {
    if (RELU) {
      asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(v[r]) : "vcc");  // not volatile: dropped where nobody reads the masks (inference)
      if (!RELU_PACKED) v[r] = fmaxf(v[r], 0.f);  // = v > 0 ? v : 0 (NaN -> 0 either way)
    }
  }
  if (RELU) mask |= (unsigned long long)bits << (t * 16);
  if (RNE) store_tile_rne(*dst, v);
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int p = 0; p < 4; p++) {
      uint32_t a_, b_;
      if (X2) {
        a_ = rne2(v[8 * s + 2 * p], v[8 * s + 2 * p + 1]); b_ = 0u;
        if (RELU_PACKED) {
          using h2 = __attribute__((ext_vector_type(2))) _Float16;
          a_ = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2, a_), h2{(_Float16)0, (_Float16)0}));
        }
      }
      else split2(v[8 * s + 2 * p], v[8 * s + 2 * p + 1], a_, b_);
      f.hi[s][p] = a_; f.lo[s][p] = b_;
    }
}
__device__ __forceinline__ void split8(const float* x, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; p++) {
    uint32_t a_, b_;
    split2(x[2 * p], x[2 * p + 1], a_, b_);
    hi[p] = a_; lo[p] = b_;
  }
}
__device__ __forceinline__ void load8(const float* p, float* x) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

// a tile's two fragments -> columns col0 .. col0+31 of sample q in the plane (base_hi, base_lo) of width W
// (registers 8s+4g..+3 = rows 16s+8g+4h..+3: lane half h owns 4 of the 8 features of octet (col0 + 16s + 8g) / 8)
__device__ __forceinline__ void store_tile(uint16_t* __restrict__ base_hi, uint16_t* __restrict__ base_lo, int W, int64_t q, int col0, const Frag& f,
                                           int h) {
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int64_t o = plane_off(q, W, (col0 >> 3) + 2 * s + g) + 4 * h;
      *reinterpret_cast<uint2*>(base_hi + o) = make_uint2(f.hi[s][2 * g], f.hi[s][2 * g + 1]);
      *reinterpret_cast<uint2*>(base_lo + o) = make_uint2(f.lo[s][2 * g], f.lo[s][2 * g + 1]);
    }
}
// the first k-step of a tile (rows 0..15) -> 16 columns
__device__ __forceinline__ void store_half_tile(uint16_t* __restrict__ base_hi, uint16_t* __restrict__ base_lo, int W, int64_t q, int col0,
                                                const u32x4& fh, const u32x4& fl, int h) {
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int64_t o = plane_off(q, W, (col0 >> 3) + g) + 4 * h;
    *reinterpret_cast<uint2*>(base_hi + o) = make_uint2(fh[2 * g], fh[2 * g + 1]);
    *reinterpret_cast<uint2*>(base_lo + o) = make_uint2(fl[2 * g], fl[2 * g + 1]);
  }
}
// a natural-order fragment (k = 8h + i) -> 16 columns: lane half h owns octet col0 / 8 + h whole
__device__ __forceinline__ void store_nat(uint16_t* __restrict__ base_hi, uint16_t* __restrict__ base_lo, int W, int64_t q, int col0,
                                          const u32x4& fh, const u32x4& fl, int h) {
  const int64_t o = plane_off(q, W, (col0 >> 3) + h);
  *reinterpret_cast<u32x4*>(base_hi + o) = fh;
  *reinterpret_cast<u32x4*>(base_lo + o) = fl;
}

// the half tile and the natural-order fragment for a one-plane buffer (PL_RNE): r = the values rounded to nearest, in the fragment's order
__device__ __forceinline__ void store_half_tile_rne(uint16_t* __restrict__ base, int W, int64_t q, int col0, const u32x4& r, int h) {
#pragma unroll
  for (int g = 0; g < 2; g++) *reinterpret_cast<uint2*>(base + plane_off(q, W, (col0 >> 3) + g) + 4 * h) = make_uint2(r[2 * g], r[2 * g + 1]);
}
__device__ __forceinline__ void store_nat_rne(uint16_t* __restrict__ base, int W, int64_t q, int col0, const u32x4& r, int h) {
  *reinterpret_cast<u32x4*>(base + plane_off(q, W, (col0 >> 3) + h)) = r;
}

// One 32-sample tile of the decoder (nerfacto_field.py:455-555): split hash features (k-steps xh/xl) -> base_network ->
// sigma, [geo | SH | appearance] -> mlp_head -> rgb.  sigma / h0 / rgb are meaningful on the h = 0 lanes; the ReLU masks
// on both halves.  PLANES (PL_*): every layer input is also written to the f16 planes at row q (global stage).
struct DecoderOut {
  float sigma, h0, r, g, b;
  unsigned long long m_a1, m_c1, m_c2;
};
// EMB_FOLDED (inference, round 6): the appearance part of head layer 1 -- W_h0[:, 31:63] emb(cam), the same 128 numbers for every sample of every ray of a
// frame (one camera per image: perssampler.py:431-433) -- already sits in the layer's bias (fold_embedding_into_bias, once per workgroup), so the layer runs
// its geo and SH k-steps only: 24 of the tile's 216 MFMAs, two operand loads + splits and eight weight-fragment reads less per 32 samples.
// X2 (inference only; 0 = the split-f16 x3 form everywhere): a mask of the layers whose ACTIVATION operand is one f16 number (rounded to nearest; the
// weights stay split), i.e. two MFMAs per product sum instead of three and no lo' half to make -- X2_B0 base layer 1 (the hash features), X2_B1 base layer 2,
// X2_H0 head layer 1 (geo | SH), X2_H1 head layer 2, X2_H2 head layer 3.  The two base layers and head layer 1's geo input feed the DENSITY (h0 -> exp);
// head layers 2 and 3 only the colour behind its sigmoid.
constexpr int X2_B0 = 1, X2_B1 = 2, X2_H0 = 4, X2_H1 = 8, X2_H2 = 16, X2_ALL = 31;
template <int PLANES, bool EMB_FOLDED = false, int X2 = 0>
__device__ __forceinline__ DecoderOut decoder_tile(const u32x4* __restrict__ Ihi, const u32x4* __restrict__ Ilo, const float* __restrict__ bias,
                                                   int lane, const u32x4 (&xh)[2], const u32x4 (&xl)[2], const float* __restrict__ sh_row,
                                                   const float* __restrict__ emb_row, uint16_t* __restrict__ planes_hi,
                                                   uint16_t* __restrict__ planes_lo, int64_t cap, int64_t q, float pc = 1.f,
                                                   const u32x4* __restrict__ sh_split = nullptr /* {hi, lo'} of this lane's 8 SH values, made by the caller once per ray */) {
  const int h = lane >> 5;
  const float* bias_h = bias + 4 * h;
  constexpr int Bb0 = 0, Bb1 = 128, Bh0 = 160, Bh1 = 288, Bh2 = 416;  // offsets of the five layers' biases (floats)
  DecoderOut out;
#define FPL(half, C) (planes_##half + cap * (C))
  {
    // Weight fragments are read one chunk ahead of their MFMAs throughout (mma3p; `w` holds the chunk about to be used).  The two narrow layers
    // (base L2: 128 -> 16, head L3: 128 -> 3) consume each 32-row tile of their input as soon as it exists -- the same k-steps in the same order on
    // their accumulator, so the same bits -- instead of after all four: three tiles of fragments (48 registers) are never alive at once, which is
    // what pays for the prefetch in the register-tight kernels.
    WFrag w = wfrag_load(Ihi, Ilo, C_B0, lane);
    // ---- base layer 1: 32 -> 128, ReLU (k-step u of the feature row: columns 16u + 8h + i), feeding base layer 2: 128 -> 16 (padded to one tile)
    unsigned long long m_a1 = 0;
    Acc2 hh;
    acc_init_bias(hh, bias_h, Bb1, 0);
#pragma unroll
    for (int ot = 0; ot < 4; ot++) {
      Acc2 a;
      acc_init_bias(a, bias_h, Bb0, ot);
      mma3p<(X2 & X2_B0) != 0>(a, w, Ihi, Ilo, C_B0 + 4 + ot, lane, xh[0], xl[0]);
      mma3p<(X2 & X2_B0) != 0>(a, w, Ihi, Ilo, C_B1 + 2 * ot, lane, xh[1], xl[1]);
      Frag a1;
      const PlaneDst pd{FPL(hi, FP_A1), 128, q, 32 * ot, h, pc, true};
      tile_to_frag<true, PLANES == PL_RNE, (X2 & X2_B1) != 0>(a, a1, m_a1, ot, &pd);
      if (PLANES == PL_SPLIT) store_tile(FPL(hi, FP_A1), FPL(lo, FP_A1), 128, q, 32 * ot, a1, h);
      mma3p<(X2 & X2_B1) != 0>(hh, w, Ihi, Ilo, C_B1 + 2 * ot + 1, lane, a1.hi[0], a1.lo[0]);
      mma3p<(X2 & X2_B1) != 0>(hh, w, Ihi, Ilo, ot < 3 ? C_B0 + ot + 1 : C_H0, lane, a1.hi[1], a1.lo[1]);
    }
    out.h0 = acc_get(hh, 0);            // row 0 = density channel: register 0 of the h=0 lanes
    out.sigma = expf(out.h0 + 1.0f);    // trunc_exp(h0 + 1), nerfacto_field.py:499
    u32x4 gh, gl;  // rows 0..15 of the tile = k-step 0 (row 0 meets zero weights)
    {
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; r++) v[r] = acc_get(hh, r);
      if (X2 & X2_H0) { rne8(v, gh); gl = u32x4{0u, 0u, 0u, 0u}; } else split8(v, gh, gl);
      if (PLANES == PL_SPLIT) store_half_tile(FPL(hi, FP_XP), FPL(lo, FP_XP), 64, q, 0, gh, gl, h);
      if (PLANES == PL_RNE) { u32x4 r; rne8(v, r, pc); store_half_tile_rne(FPL(hi, FP_XP), 64, q, 0, r, h); }
    }
    // ---- head layer 1: [geo | SH | appearance] -> 128, ReLU
    u32x4 sh_h, sh_l, e_h[2], e_l[2];
    {
      float x[8];
      if (PLANES == PL_NONE && sh_split) { sh_h = sh_split[0]; sh_l = sh_split[1]; }
      else { load8(sh_row + 8 * h, x); if (X2 & X2_H0) { rne8(x, sh_h); sh_l = u32x4{0u, 0u, 0u, 0u}; } else split8(x, sh_h, sh_l); }
      if (PLANES == PL_RNE) { u32x4 r; rne8(x, r, pc); store_nat_rne(FPL(hi, FP_XP), 64, q, 16, r, h); }
      if (EMB_FOLDED) {
      } else if (emb_row) {
        const float* ep = emb_row;
#pragma unroll
        for (int u = 0; u < 2; u++) {
          load8(ep + 16 * u + 8 * h, x); split8(x, e_h[u], e_l[u]);
          if (PLANES == PL_RNE) { u32x4 r; rne8(x, r, pc); store_nat_rne(FPL(hi, FP_XP), 64, q, 32 + 16 * u, r, h); }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
          for (int p = 0; p < 4; p++) { e_h[u][p] = 0u; e_l[u][p] = 0u; }
          if (PLANES == PL_RNE) store_nat_rne(FPL(hi, FP_XP), 64, q, 32 + 16 * u, e_h[u], h);
        }
      }
      if (PLANES == PL_SPLIT) {
        store_nat(FPL(hi, FP_XP), FPL(lo, FP_XP), 64, q, 16, sh_h, sh_l, h);
        store_nat(FPL(hi, FP_XP), FPL(lo, FP_XP), 64, q, 32, e_h[0], e_l[0], h);
        store_nat(FPL(hi, FP_XP), FPL(lo, FP_XP), 64, q, 48, e_h[1], e_l[1], h);
      }
    }
    Frag c1[4];
    unsigned long long m_c1 = 0;
#pragma unroll
    for (int ot = 0; ot < 4; ot++) {
      Acc2 a;
      acc_init_bias(a, bias_h, Bh0, ot);
      constexpr bool x2h0 = (X2 & X2_H0) != 0;
      mma3p<x2h0>(a, w, Ihi, Ilo, C_H0 + 1 * 4 + ot, lane, gh, gl);
      if (EMB_FOLDED) {
        mma3p<x2h0>(a, w, Ihi, Ilo, ot < 3 ? C_H0 + ot + 1 : C_H1, lane, sh_h, sh_l);
      } else {
        mma3p<x2h0>(a, w, Ihi, Ilo, C_H0 + 2 * 4 + ot, lane, sh_h, sh_l);
        mma3p<x2h0>(a, w, Ihi, Ilo, C_H0 + 3 * 4 + ot, lane, e_h[0], e_l[0]);
        mma3p<x2h0>(a, w, Ihi, Ilo, ot < 3 ? C_H0 + ot + 1 : C_H1, lane, e_h[1], e_l[1]);
      }
      const PlaneDst pd{FPL(hi, FP_C1), 128, q, 32 * ot, h, pc, true};
      tile_to_frag<true, PLANES == PL_RNE, (X2 & X2_H1) != 0>(a, c1[ot], m_c1, ot, &pd);
      if (PLANES == PL_SPLIT) store_tile(FPL(hi, FP_C1), FPL(lo, FP_C1), 128, q, 32 * ot, c1[ot], h);
    }
    // ---- head layer 2: 128 -> 128, ReLU, feeding head layer 3: 128 -> 3 (padded), sigmoid
    unsigned long long m_c2 = 0;
    Acc2 o3;
    acc_init_bias(o3, bias_h, Bh2, 0);
#pragma unroll
    for (int ot = 0; ot < 4; ot++) {
      Acc2 a;
      acc_init_bias(a, bias_h, Bh1, ot);
#pragma unroll
      for (int u = 0; u < 8; u++) mma3p<(X2 & X2_H1) != 0>(a, w, Ihi, Ilo, u < 7 ? C_H1 + (u + 1) * 4 + ot : C_H2 + 2 * ot, lane, c1[u >> 1].hi[u & 1], c1[u >> 1].lo[u & 1]);
      Frag c2;
      const PlaneDst pd{FPL(hi, FP_C2), 128, q, 32 * ot, h, pc, true};
      tile_to_frag<true, PLANES == PL_RNE, (X2 & X2_H2) != 0>(a, c2, m_c2, ot, &pd);
      if (PLANES == PL_SPLIT) store_tile(FPL(hi, FP_C2), FPL(lo, FP_C2), 128, q, 32 * ot, c2, h);
      mma3p<(X2 & X2_H2) != 0>(o3, w, Ihi, Ilo, C_H2 + 2 * ot + 1, lane, c2.hi[0], c2.lo[0]);
      mma3p<(X2 & X2_H2) != 0>(o3, w, Ihi, Ilo, ot < 3 ? C_H1 + ot + 1 : -1, lane, c2.hi[1], c2.lo[1]);
    }
    out.r = 1.f / (1.f + expf(-acc_get(o3, 0)));  // rows 0..2 = registers 0..2 of the h=0 lanes
    out.g = 1.f / (1.f + expf(-acc_get(o3, 1)));
    out.b = 1.f / (1.f + expf(-acc_get(o3, 2)));
    out.m_a1 = m_a1; out.m_c1 = m_c1; out.m_c2 = m_c2;
    return out;
  }
#undef FPL
}

// bias_h0[r] += sum_k W_h0[r][31 + k] emb[k] for the 128 rows of head layer 1, in the workgroup's LDS image (threads 0 .. 127, one row each; the caller
// puts __syncthreads() on both sides).  The weights are read back from the image's split-f16 fragments -- chunk C_H0 + 4 u + ot holds k-step u of output
// tile ot, lane (h, j) of it row 32 ot + j, inputs 16 u + 8 h .. + 7, as hi + lo' 2^-11 -- and summed in f32 in a fixed order: hi * e, then lo' * e scaled,
// the two products the three-MFMA form keeps of (hi + lo')(e_hi + e_lo') plus the e_lo' hi term it also keeps; only rounding order differs (1e-7 relative).
__device__ __forceinline__ void fold_embedding_into_bias(u32x4* __restrict__ L, const float* __restrict__ emb_row) {
  if (threadIdx.x >= 128) return;
  const int r = threadIdx.x, ot = r >> 5, j = r & 31;
  float* bias = reinterpret_cast<float*>(L + 2 * FWD_CHUNKS * 64);
  constexpr int Bh0 = 160;
  float acc = bias[Bh0 + r];
#pragma unroll
  for (int u = 2; u < 4; u++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int e = (C_H0 + 4 * u + ot) * 64 + 32 * h + j;
      const h8 whi = __builtin_bit_cast(h8, L[e]), wlo = __builtin_bit_cast(h8, L[FWD_CHUNKS * 64 + e]);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float w = __builtin_fmaf((float)wlo[i], 0x1p-11f, (float)whi[i]);
        acc = __builtin_fmaf(w, emb_row[16 * (u - 2) + 8 * h + i], acc);
      }
    }
  bias[Bh0 + r] = acc;
}

template <bool RELU_MASK, bool RNE = false>
__device__ __forceinline__ void grad_tile_to_frag(const Acc2& a, Frag& f, unsigned long long mask, int t, const PlaneDst* dst = nullptr) {
  float v[16];
  acc_get16(a, v);
  const int bits = (int)(uint32_t)(mask >> (t * 16));  // this tile's 16 ReLU bits
#pragma unroll
  for (int r = 0; r < 16; r++) {
    // ReLU': and with the bit stretched to a word (v_bfe_i32 + v_and instead of shift / and / compare / select on 64 bits)
    if (RELU_MASK) v[r] = __int_as_float(__float_as_int(v[r]) & ((bits << (31 - r)) >> 31));
  }
  if (RNE) store_tile_rne(*dst, v);
#pragma unroll
  for (int s = 0; s < 2; s++) split8(v + 8 * s, f.hi[s], f.lo[s]);
}

// Sum v over the lanes [lane, run_end] of this 32-lane half (runs = consecutive samples of one ray).
__device__ __forceinline__ float half_run_sum(float v, int j, int run_end) {
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const float u = __shfl_down(v, off, 32);
    if (j + off <= run_end) v += u;
  }
  return v;
}


}  // namespace
