// fp32 products on the bf16 matrix pipe: an fp32 number as the exact sum of three bf16 pieces.
//
// x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m), round to nearest even each time: 8 + 8 + 8 significand bits, the
// subtractions are exact, the sum of the pieces is x (or x rounded in its 25th bit).  Of the nine piece products of x w the six of
// weight >= 2^-16 are issued by the callers - (l,h) (m,m) (m,h) (h,l) (h,m) (h,h), in that order, every one exact as an input of
// the fp32 accumulator; what is dropped is below 2^-23 |x w|, one fp32 rounding.  Used by mlp2_split_kernel (gemm.hip) and
// gram_split_kernel (kernel_reg.hip); measured against fp64 both are closer than the k-ordered fp32 fma chain.
// Non-finite inputs: a NaN stays a NaN (every piece of it is one); an infinity - and a finite |x| > 3.3895e38, which rounds to
// the bf16 infinity - gives NaN where the fp32 chain gives +-inf (x - x_h = inf - inf).  Features and weights are finite;
// WDG_MLP2_SPLIT=0 / WDG_GRAM_SPLIT=0 select the chain for inputs that are not.
#pragma once
#include "wdg_common.h"

namespace wdg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // (v_cvt_pk_bf16_f32: round to nearest even, a in the low half)
    const bf16x2_t v = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}

// (the subtractions as single v_sub_f32: left to itself the compiler pairs them into v_pk_add_f32, which costs more than two
// plain instructions beside MFMAs - MI355X_MICROARCH "packed f32 VALU ... an anti-lever beside MFMAs")
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// two floats -> their three bf16 pieces, packed pairwise (x0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = pack_bf16(x0, x1);
    const float r0 = sub_f32(x0, __uint_as_float(h << 16)), r1 = sub_f32(x1, __uint_as_float(h & 0xffff0000u));
    m = pack_bf16(r0, r1);
    const float s0 = sub_f32(r0, __uint_as_float(m << 16)), s1 = sub_f32(r1, __uint_as_float(m & 0xffff0000u));
    l = pack_bf16(s0, s1);
}

__device__ __forceinline__ bf16x8_t as_frag(const u32x4_t &v) { return __builtin_bit_cast(bf16x8_t, v); }

}  // namespace wdg
