// derive.hip -- derived input variables of the loader (climsim_utils/data_utils.py:654-697, get_xrdata): variables a
// variable set asks for that are not stored in the files are computed from the stored state on read:
//   state_rh      = state_q0001 / qvs,  qvs = Rd esat / (Rv pmid),  esat = omega eliq(T) + (1 - omega) eice(T),
//                   omega = clip((T - 253.16) / 20, 0, 1)                                            (:662-673)
//   liq_partition = omega                                                                               (:684-690)
//   state_qn      = state_q0002 + state_q0003   (also the *_prvphy pairs: call with those arrays)     (:692-707)
// eliq / eice (:19-43) are 8th-order polynomials that numpy evaluates in float64 (np.polyval with float64 coefficients);
// the kernel does the same: fp32 in, float64 Horner, fp32 out.  One pass, HBM-bound: 12-16 bytes read, 4-12 written per cell.
#include "common.h"

namespace {

__device__ __forceinline__ double polyval9(const double *a, double x)
{
    double o = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o = o * x + a[i];      // numpy.polyval: Horner, y = y * x + p[i]
    return o;
}

// The reference applies these to float32 arrays: `T - T0` is then formed in float32 (python-scalar operand) and everything
// downstream of it in float64 (float64 coefficient arrays), which is what `x` = (double)(float)(T - 273.16f) reproduces.
__device__ __forceinline__ double eliq_d(float T)
{
    const double a[9] = {-0.976195544e-15, -0.952447341e-13, 0.640689451e-10, 0.206739458e-7, 0.302950461e-5,
                         0.264847430e-3, 0.142986287e-1, 0.443987641, 6.11239921};
    return 100.0 * polyval9(a, fmax(-80.0, (double)(T - 273.16f)));
}

__device__ __forceinline__ double eice_d(float T)
{
    const double a[9] = {0.252751365e-14, 0.146898966e-11, 0.385852041e-9, 0.602588177e-7, 0.615021634e-5,
                         0.420895665e-3, 0.188439774e-1, 0.503160820, 6.11147274};
    if ((double)T > 273.15) return eliq_d(T);
    if ((double)T > 185.0) return 100.0 * polyval9(a, (double)(T - 273.16f));
    const double t = fmax(-100.0, (double)(T - 273.16f));
    return 100.0 * (0.00763685 + t * (0.000151069 + t * 7.48215e-07));
}

__global__ __launch_bounds__(256) void derive_kernel(long n, const float *__restrict__ T, const float *__restrict__ q1,
                                                     const float *__restrict__ pmid, const float *__restrict__ q2,
                                                     const float *__restrict__ q3, float *__restrict__ rh,
                                                     float *__restrict__ liq, float *__restrict__ qn)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (rh || liq) {
            const float t = T[i];
            // the reference forms omega in the array's own precision (float32: `tair - T00` with a python scalar)
            const float om = fminf(1.0f, fmaxf(0.0f, (t - 253.16f) / (273.16f - 253.16f)));
            // a missing cell stays missing: np.maximum / np.minimum propagate NaN (fmaxf / fminf drop it), and for +-inf the masked
            // polynomial branches of eice give 0 * inf = NaN upstream
            const bool finite = fabsf(t) <= 3.402823466e38f;
            if (liq) liq[i] = (t != t) ? t : om;
            if (rh && !finite) rh[i] = __builtin_nanf("");
            else if (rh) {
                const double esat = (double)om * eliq_d(t) + (double)(1.0f - om) * eice_d(t);
                const double qvs = (287.0 * esat) / (double)(461.0f * pmid[i]);      // `Rv * pmid` stays float32 upstream
                rh[i] = (float)((double)q1[i] / qvs);
            }
        }
        if (qn) qn[i] = q2[i] + q3[i];
    }
}

}  // namespace

extern "C" int csa_derive_inputs(long n, const float *state_t, const float *state_q0001, const float *state_pmid,
                                 const float *q2, const float *q3, float *state_rh, float *liq_partition, float *state_qn,
                                 void *stream)
{
    if (n <= 0 || (!state_rh && !liq_partition && !state_qn)) { csa_set_error_msg("csa_derive_inputs: nothing to do"); return CSA_ERR_ARG; }
    if ((state_rh || liq_partition) && !state_t) { csa_set_error_msg("csa_derive_inputs: state_t required"); return CSA_ERR_ARG; }
    if (state_rh && (!state_q0001 || !state_pmid)) { csa_set_error_msg("csa_derive_inputs: state_rh needs state_q0001 and state_pmid"); return CSA_ERR_ARG; }
    if (state_qn && (!q2 || !q3)) { csa_set_error_msg("csa_derive_inputs: state_qn needs both cloud species"); return CSA_ERR_ARG; }
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(derive_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, n, state_t,
                       state_q0001, state_pmid, q2, q3, state_rh, liq_partition, state_qn);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
