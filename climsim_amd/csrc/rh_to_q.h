// rh_to_q.h -- device functions shared by prep.hip (q as a level input) and head.hip (mp_mode -2 reads q_old).
#pragma once

// Specific humidity from relative humidity (rnn/utils.py:134-180, relative_to_specific_humidity_torch):
// 8th-order Horner polynomials for the saturation vapour pressure over liquid / ice, blended by
// omega = clamp((T-253.16)/20, 0, 1); q = rh * Rd*esat / (Rv*p).
// unfusable multiply / add (inline asm: hipcc contracts even __fmul_rn + __fadd_rn into v_fma)
__device__ __forceinline__ float mul_nofma(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add_nofma(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float prep_polyval9(const float *a, float x)
{
    // separate multiply and add, exactly as torch evaluates `out * x + c` (contraction is switched off for
    // these two functions): near -87 C the fp32 Horner sum cancels to 1e-6 of its terms, the reference's own
    // value is then ~17 % from the exact polynomial and an FMA evaluation lands ~16 % away from the reference
    float o = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) o = add_nofma(mul_nofma(o, x), a[i]);
    return o;
}
__device__ __forceinline__ float prep_rh_to_q(float rh, float T, float p)
{
    const float a_liq[9] = {-0.976195544e-15f, -0.952447341e-13f, 0.640689451e-10f, 0.206739458e-7f, 0.302950461e-5f,
                            0.264847430e-3f, 0.142986287e-1f, 0.443987641f, 6.11239921f};
    const float a_ice[9] = {0.252751365e-14f, 0.146898966e-11f, 0.385852041e-9f, 0.602588177e-7f, 0.615021634e-5f,
                            0.420895665e-3f, 0.188439774e-1f, 0.503160820f, 6.11147274f};
    const float T0 = 273.16f;
    const float eliq = 100.0f * prep_polyval9(a_liq, fmaxf(T - T0, -80.0f));
    float eice;
    if (T > 273.15f) eice = eliq;
    else if (T > 185.0f) eice = 100.0f * prep_polyval9(a_ice, T - T0);
    else {
        const float tmp = fmaxf(T - T0, -100.0f);
        eice = 100.0f * add_nofma(0.00763685f, mul_nofma(tmp, add_nofma(0.000151069f, mul_nofma(tmp, 7.48215e-07f))));
    }
    float omega = (T - 253.16f) / 20.0f;
    omega = fminf(fmaxf(omega, 0.0f), 1.0f);
    const float esat = add_nofma(mul_nofma(omega, eliq), mul_nofma(1.0f - omega, eice));
    return rh * ((287.0f * esat) / (461.0f * p));
}

