// pt_math.hpp — the numeric contract on the device.
//
// Every float32 operation of the render path is an IEEE-754 binary32 +,-,*,/,sqrt or an explicit
// fused multiply-add in a fixed order, so that results do not depend on the compiler: built with
// -ffp-contract=off (only the explicit __builtin_fmaf calls fuse) and correctly rounded divide /
// sqrt (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).  The hardware approximations
// (v_sin_f32, v_log_f32, v_rcp_f32, ...) are deliberately not used: they cannot be reproduced
// off-chip.  GLSL built-ins restated here: sin cos log exp atan asin normalize dot cross length
// distance reflect refract mix min max (frag.glsl, used at :235-283, :351-419, :696-708, :726-882).
// Polynomial coefficients: Cephes single precision (public domain); log: fdlibm e_logf reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_DEV __device__ __forceinline__ static

namespace pm {


PM_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
PM_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
PM_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PM_DEV bool isnan_(float x) { return x != x; }

// ---- elementary functions ---------------------------------------------------------------------

// Cody-Waite reduction of x to r in [-pi/4, pi/4], quadrant q (valid for |x| < ~1e5).
PM_DEV float reduce_pio2(float x, int& q) {
    const float TWO_OVER_PI = u2f(0x3f22f983u);
    const float PIO2_HI = u2f(0x3fc90fdbu), PIO2_MID = u2f(0xb33bbd2eu), PIO2_LO = u2f(0xa6f72cedu);
    float k = __builtin_rintf(x * TWO_OVER_PI);
    q = (int)k;
    float r = fma_(-k, PIO2_HI, x);
    r = fma_(-k, PIO2_MID, r);
    r = fma_(-k, PIO2_LO, r);
    return r;
}
PM_DEV float sin_poly(float r) {
    float z = r * r;
    float p = fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return fma_(p * z, r, r);
}
PM_DEV float cos_poly(float r) {
    float z = r * r;
    float p = fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return fma_(p * z, z, fma_(-0.5f, z, 1.0f));
}
PM_DEV float sin_(float x) {
    int q; float r = reduce_pio2(x, q);
    float s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
PM_DEV float cos_(float x) {
    int q; float r = reduce_pio2(x, q);
    float c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

PM_DEV float log_(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0) return -__builtin_inff();       // log(+-0) = -inf
        if (ix >> 31) return __builtin_nanf("");                    // log(negative) = NaN
        k -= 25; x *= 33554432.0f; ix = f2u(x);      // subnormal: scale by 2^25
    } else if (ix >= 0x7f800000u) {
        return x;                                    // +inf, NaN
    } else if (ix == 0x3f800000u) {
        return 0.0f;
    }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = u2f(ix);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

PM_DEV float exp_(float x) {
    const float LOG2E = u2f(0x3fb8aa3bu), LN2_HI = u2f(0x3f317218u), LN2_LO = u2f(0xb102e308u);
    if (isnan_(x)) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.97208404541015625f) return 0.0f;
    float k = __builtin_rintf(x * LOG2E);
    float r = fma_(-k, LN2_HI, x);
    r = fma_(-k, LN2_LO, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    p = fma_(p, r * r, r) + 1.0f;
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;                   // 2^k = 2^k1 * 2^k2, both normal
    return (p * u2f((uint32_t)(k1 + 127) << 23)) * u2f((uint32_t)(k2 + 127) << 23);
}

PM_DEV float atan_(float x) {
    const float PIO2 = u2f(0x3fc90fdbu), PIO4 = u2f(0x3f490fdbu);
    float t = __builtin_fabsf(x);
    float y0;
    if (t > 2.414213562373095f) { y0 = PIO2; t = -(1.0f / t); }
    else if (t > 0.4142135623730950f) { y0 = PIO4; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float p = fma_(fma_(fma_(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    float y = y0 + fma_(p * z, t, t);
    return (f2u(x) >> 31) ? -y : y;
}
// GLSL atan(y, x)
PM_DEV float atan2_(float y, float x) {
    const float PI = u2f(0x40490fdbu), PIO2 = u2f(0x3fc90fdbu);
    if (x > 0.0f) return atan_(y / x);
    if (x < 0.0f) { float a = atan_(y / x); return (y >= 0.0f) ? a + PI : a - PI; }
    if (x == 0.0f) { if (y > 0.0f) return PIO2; if (y < 0.0f) return -PIO2; if (y == 0.0f) return 0.0f; }
    return __builtin_nanf("");
}
PM_DEV float asin_(float x) {
    const float PIO2 = u2f(0x3fc90fdbu);
    float a = __builtin_fabsf(x);
    if (!(a <= 1.0f)) return __builtin_nanf("");                    // |x| > 1 or NaN
    if (a < 1.0e-4f) return x;
    float z, xx; bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); xx = __builtin_sqrtf(z); } else { xx = a; z = xx * xx; }
    float p = fma_(fma_(fma_(fma_(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    float r = fma_(p * z, xx, xx);
    if (big) { r = r + r; r = PIO2 - r; }
    return (f2u(x) >> 31) ? -r : r;
}

// ---- vec3 ---------------------------------------------------------------------------------------
struct vec3 { float x, y, z; };
PM_DEV vec3 v3(float a) { return {a, a, a}; }
PM_DEV vec3 v3(float a, float b, float c) { return {a, b, c}; }
PM_DEV vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PM_DEV vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PM_DEV vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PM_DEV vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PM_DEV vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
// a*s + b, one fused multiply-add per component (ray point evaluation o + t*d)
PM_DEV vec3 madd(vec3 a, float s, vec3 b) { return {fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)}; }
PM_DEV float dot(vec3 a, vec3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PM_DEV vec3 cross(vec3 a, vec3 b) {
    return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
PM_DEV float length(vec3 a) { return __builtin_sqrtf(dot(a, a)); }
PM_DEV float distance(vec3 a, vec3 b) { return length(a - b); }
PM_DEV vec3 normalize(vec3 a) { float inv = 1.0f / __builtin_sqrtf(dot(a, a)); return a * inv; }
PM_DEV vec3 reflect(vec3 I, vec3 N) { float k = 2.0f * dot(N, I); return madd(N, -k, I); }
PM_DEV vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f);
    float s = fma_(eta, d, __builtin_sqrtf(k));
    return madd(N, -s, I * eta);
}
PM_DEV vec3 mix(vec3 x, vec3 y, float a) { return madd(y, a, x * (1.0f - a)); }
// exp() applied component-wise
PM_DEV vec3 exp3(vec3 a) { return {exp_(a.x), exp_(a.y), exp_(a.z)}; }

// ---- the opt-in relaxed contract (pt_set_option: numeric_contract = 1, bench.py --contract fast) ------------------------------
// The exact contract above prices Box-Muller at ~90 instructions per Gaussian and every divide / sqrt / normalize at 11-13.  The relaxed
// one keeps the RNG, the draw counts and every branch of the shader as written and only evaluates the continuous functions with the
// hardware's units (v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_log_f32, v_cos_f32: 1 ulp; v_cos ~1e-6 absolute).  Results are no longer
// bit-identical to the oracle; the bar is north_star's per-pixel RMSE <= 1e-3, asserted by tests/test_gpu_parity.py::test_fast_contract_*.
// Not part of oracle/glsl_math.h: the oracle only ever implements the exact contract.
template <bool FAST> PM_DEV float rcpT(float x) { if (FAST) return __builtin_amdgcn_rcpf(x); return 1.0f / x; }
template <bool FAST> PM_DEV float divT(float a, float b) { if (FAST) return a * __builtin_amdgcn_rcpf(b); return a / b; }
template <bool FAST> PM_DEV float sqrtT(float x) { if (FAST) return __builtin_amdgcn_sqrtf(x); return __builtin_sqrtf(x); }
template <bool FAST> PM_DEV vec3 normalizeT(vec3 a) { if (FAST) return a * __builtin_amdgcn_rsqf(dot(a, a)); return normalize(a); }
template <bool FAST> PM_DEV float lengthT(vec3 a) { return sqrtT<FAST>(dot(a, a)); }
// rho * cos(theta) of randValNormalDist (frag.glsl:696-701) from the two uniform draws
template <bool FAST> PM_DEV float boxMullerT(float u1, float u2) {
    if (FAST) {
        const float rho = __builtin_amdgcn_sqrtf(-2.0f * (__builtin_amdgcn_logf(u2) * 0.69314718056f));      // v_log_f32 is log2
        return rho * __builtin_amdgcn_cosf(u1 * (2.0f * 3.1415926f * 0.15915494309f));                      // v_cos_f32 takes revolutions
    }
    const float theta = 2.0f * 3.1415926f * u1;
    const float rho = __builtin_sqrtf(-2.0f * log_(u2));
    return rho * cos_(theta);
}
template <bool FAST> PM_DEV vec3 refractT(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f);
    float s = fma_(eta, d, sqrtT<FAST>(k));
    return madd(N, -s, I * eta);
}

// min/max as used by rayBox (frag.glsl:412-415).  GLSL leaves NaN behaviour undefined; the
// contract is IEEE minNum/maxNum (a NaN operand is ignored), which is what GPU min/max
// instructions implement.  Only comparisons consume the results, so the sign of zero is moot.
PM_DEV float minnum(float a, float b) { return __builtin_fminf(a, b); }   // v_min_f32 / v_min3_f32
PM_DEV float maxnum(float a, float b) { return __builtin_fmaxf(a, b); }   // v_max_f32 / v_max3_f32

}  // namespace pm
