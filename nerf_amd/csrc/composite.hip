// composite.hip -- output activation, alpha compositing + loss, and the compositing gradient.
//
// Replaces the reference's three CUDA kernels:
//   activation_kernel                  nerf.c:7-19   (launch train.c:192-195, nerf.c:287-290)
//   volume_rendering_and_loss_kernel   nerf.c:21-61  (launch train.c:204-207)
//   volume_rendering_gradient_kernel   nerf.c:63-127 (launch train.c:210-212)
//
// One 64-lane wavefront owns one ray.  Lane l holds the SPL = ceil(S/64) consecutive samples
// [l*SPL, (l+1)*SPL) in registers, so a ray's S samples are read with one coalesced sweep.
//   forward : transmittance = exclusive product scan of (1 - alpha) across the wavefront
//             (lane-local running product, then a 6-step __shfl_up scan of the lane products);
//             a sample contributes iff the transmittance in front of it is >= T_STOP, which is
//             the reference's "accumulate, then break when T < 0.01" rule (T never increases);
//             the pixel sum is a butterfly reduction.  The scan associates the product differently
//             from the reference's left-to-right loop (nerf.c:42), so a transmittance within a few
//             ulps of 0.01 could fall on the other side of the stop: a ray on which any scanned
//             value lies inside the provable error band around T_STOP (guard_band(), about one ray
//             in 10^5) is multiplied up again in the reference's order and decided by that.
//   backward: the reference's O(S^3) density-gradient loop equals the reverse recurrence
//             R_{S-1} = 0, R_s = (1-a_s)(a_{s+1} q_{s+1} + R_{s+1}), G_s = da_s (T_s q_s - R_s)
//             with q_s = sum_c err_c rgb_{s,c}; it is evaluated as a reverse scan under
//             composition of the affine maps x -> (1-a_s) x + (1-a_s) a_{s+1} q_{s+1}.
//             No early stop here (nerf.c:77-87 has none), hence no decision and no band.
// Both kernels move 16 B per sample in (+16 B out for the gradient); what they cost beyond that is
// the exponential behind each alpha and the three sigmoids per sample.
//
// SPACING = true is an opt-in quadrature that is NOT the reference's: alpha_s = 1 - exp(-dens_s delta_s)
// with the true spacing delta_s = t_{s+1} - t_s of the (possibly jittered) sample depths,
// delta_{S-1} = NERF_LAST_DELTA, no early stop, and the analytic gradient
//   dL/ddens_s = delta_s ((1 - alpha_s) T_s q_s - sum_{t>s} w_t q_t)
// whose suffix sum is a reverse additive scan across the wavefront.
#include "common.hpp"
#include "../../include/nerf_spec.h"

namespace nerf {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// e^x to about 1.5 ulp: x log2(e) split into a rounded integer and a remainder carried in two
// pieces, the hardware exp2 on the remainder, ldexp for the integer.  Overflow and underflow come out
// of the ldexp as inf and 0; no range tests.
__device__ __forceinline__ float expf_fast(float x) {
    const float t = x * 1.44269502e+00f;
    const float k = __builtin_rintf(t);
    float f = fmaf(x, 1.44269502e+00f, -k);  // the rounding error of t is recovered by the fma
    f = fmaf(x, 1.92596299e-08f, f);         // log2(e) - float(log2(e))
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)k);
}

// 1 / (1 + expf(-x)) (nerf.c:15-17).  The division is a reciprocal and one Newton step (within an ulp
// of the IEEE quotient; the denominator is >= 1) instead of the ten-instruction IEEE sequence.  The
// exponent is kept inside [-104, 88]: above 88 the denominator would overflow (sigmoid(-88) = 6e-39
// either way); below -104 expf is zero in fp32 anyway, and expf_fast's integer part would leave the
// range of an int for huge positive x (+inf, 1e30: the reference returns 1.0 there, nerf.c:15-17, and
// so does this).  Absolute error below 1e-7.
__device__ __forceinline__ float sigmoid_ref(float x) {
    const float d = 1.0f + expf_fast(fmaxf(fminf(-x, 88.0f), -104.0f));
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}

// expf rounded correctly, for x <= 0: alpha = 1 - expf(-0.01 * density) cancels for small densities,
// so a one-ulp difference in expf moves alpha by 6e-8 whatever its size -- up to 1e-3 RELATIVE on the
// weights of a thin medium.  The correctly rounded value is what a host libm returns (glibc's expf
// stays within 0.502 ulp, i.e. it is the correctly rounded value except about once in 10^3 arguments
// that lie next to a rounding boundary).  Evaluated in double precision: x = k ln2 + r with
// |r| <= ln2/2, a degree-12 Taylor polynomial of exp(r) (truncation r^13/13! < 2e-16 relative, the
// size of one double rounding: the float rounding differs from the exact one for about one argument
// in 10^8), scaled by 2^k and rounded to float once.  20 double-precision instructions, half of what
// the general-purpose exp() spends on ranges that cannot occur here.
__device__ __forceinline__ float expf_cr(float xf) {
    const double x = (double)fmaxf(xf, -104.0f);  // exp(-104) < 2^-150 rounds to zero: so does everything below (and -inf)
    const double kd = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(kd, -6.93147180369123816490e-01, x);  // ln2 split so that kd * hi is exact
    r = __builtin_fma(kd, -1.90821492927058770002e-10, r);
    double p = 2.0876756987868098979e-09;          // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);  // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);  // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);  // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);  // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);  // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);  // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);  // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);  // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);  // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)kd);
}

// Half-width of the band around T_STOP inside which a scanned transmittance does not settle the
// reference's stop decision.  Any way of multiplying up p factors in fp32 takes p - 1 roundings of
// relative size <= 2^-24 each, so two differently associated products of the same <= S factors differ
// by less than 2 S 2^-24 relative; 1.5 covers the second-order terms.  Outside the band the scanned
// value and the reference's running product lie on the same side of T_STOP.
__device__ __forceinline__ float guard_band(int S) { return NERF_T_STOP * (3.0f * (float)S) * 5.9604645e-8f; }

// Lane-crossing through the DPP path of the vector ALU (no LDS traffic, no index arithmetic): lane l
// receives src from the lane the control word names, or keeps `old` when that lane lies outside its
// row of 16 (or outside the masked rows / banks).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, ROW_MASK, BANK_MASK, false));
}
enum { DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141,
       DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_QUAD_1032 = 0xb1, DPP_QUAD_2301 = 0x4e };

// exclusive product scan over the 64 lanes of a wavefront: inclusive scans inside the four rows of
// 16 lanes, each row's total passed on to the rows after it, then one lane to the right
__device__ __forceinline__ float wave_exclusive_product(float x, int lane) {
    (void)lane;
    float v = x;
    v *= dpp<DPP_ROW_SHR + 1>(1.0f, v);
    v *= dpp<DPP_ROW_SHR + 2>(1.0f, v);
    v *= dpp<DPP_ROW_SHR + 4>(1.0f, v);
    v *= dpp<DPP_ROW_SHR + 8>(1.0f, v);
    v *= dpp<DPP_ROW_BCAST15, 0xa>(1.0f, v);  // last lane of rows 0 and 2 into every lane of rows 1 and 3
    v *= dpp<DPP_ROW_BCAST31, 0xc>(1.0f, v);  // lane 31 into every lane of rows 2 and 3
    return dpp<DPP_WAVE_SHR1>(1.0f, v);       // lane 0 keeps 1
}

// Transmittance in front of this lane's first sample, multiplied up in the REFERENCE's order: one
// running product over all samples, left to right (nerf.c:42).  Every lane walks the same chain (the
// factors come by wave-uniform lane reads), so the value each lane keeps is bit for bit what a
// sequential loop holds at that sample -- and with it the early-stop decision (nerf.c:43).  S
// dependent multiplies: only the rare rays inside the guard band take this path.
template <int SPL>
__device__ __forceinline__ float sequential_transmittance(const float (&alpha)[SPL], int lane) {
    float T = 1.0f, mine = 1.0f;
    for (int l = 0; l < 64; l++) {
        if (lane == l) mine = T;
#pragma unroll
        for (int i = 0; i < SPL; i++) T *= (1.0f - __shfl(alpha[i], l, 64));
    }
    return mine;
}

// sum over the wavefront, returned in every lane: butterflies inside each row of 16, row totals
// chained into lane 63, one scalar read
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp<DPP_QUAD_1032>(0.0f, x);
    x += dpp<DPP_QUAD_2301>(0.0f, x);
    x += dpp<DPP_ROW_HALF_MIRROR>(0.0f, x);
    x += dpp<DPP_ROW_MIRROR>(0.0f, x);
    x += dpp<DPP_ROW_BCAST15, 0xa>(0.0f, x);
    x += dpp<DPP_ROW_BCAST31, 0xc>(0.0f, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// exclusive suffix sum over the 64 lanes: lane l receives sum of x over lanes > l
__device__ __forceinline__ float wave_exclusive_suffix_sum(float x, int lane) {
    float incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float y = __shfl_down(incl, d, 64);
        if (lane + d < 64) incl += y;
    }
    const float ex = __shfl_down(incl, 1, 64);
    return lane == 63 ? 0.0f : ex;
}

// depth of sample s exactly as the encoder places it (nerf.c:187 with an optional offset in units of
// one spacing), and the spacing the opt-in quadrature gives that sample
__device__ __forceinline__ float sample_depth_of(int s, int S, float near_plane, float far_plane, const float* off) {
    const float fs = off ? __fadd_rn((float)s, off[s]) : (float)s;
    return __fadd_rn(near_plane, __fdiv_rn(__fmul_rn(__fsub_rn(far_plane, near_plane), fs), (float)(S - 1)));
}
__device__ __forceinline__ float sample_spacing(int s, int S, float near_plane, float far_plane, const float* off) {
    if (s == S - 1) return NERF_LAST_DELTA;
    return __fsub_rn(sample_depth_of(s + 1, S, near_plane, far_plane, off), sample_depth_of(s, S, near_plane, far_plane, off));
}

// One sample as it lies in memory: (density, r, g, b).  RAW: the MLP's raw output row; otherwise
// dens[] / rgb[].  The loads of all of a lane's samples are issued before anything is computed on them.
template <bool RAW>
__device__ __forceinline__ f32x4 fetch_sample(const float* __restrict__ raw, int out_dim, const float* __restrict__ dens,
                                              const float* __restrict__ rgb, size_t p) {
    if (RAW) {
        if (out_dim == 4) return *reinterpret_cast<const f32x4*>(raw + p * 4);
        const float* y = raw + p * out_dim;
        return f32x4{y[0], y[1], y[2], y[3]};
    }
    return f32x4{dens[p], rgb[p * 3 + 0], rgb[p * 3 + 1], rgb[p * 3 + 2]};
}
// RAW: the output activation (ReLU on channel 0, sigmoid on 1..3, nerf.c:13-17)
template <bool RAW>
__device__ __forceinline__ void activate_sample(const f32x4 y, float& d, float (&c)[3]) {
    if (RAW) {
        d = fmaxf(0.0f, y[0]);
        c[0] = sigmoid_ref(y[1]); c[1] = sigmoid_ref(y[2]); c[2] = sigmoid_ref(y[3]);
    } else {
        d = y[0];
        c[0] = y[1]; c[1] = y[2]; c[2] = y[3];
    }
}

struct SpacingArgs {
    const float* offsets;  // [R][S] or null
    float near_plane, far_plane;
};

// One ray's forward pass on one wavefront.  WANT_E: the pixel error is also handed back in every lane
// (e[c] = pixel[c] - target[c], the value lanes 0..2 store), for a gradient that follows in the same kernel.
template <int SPL, bool RAW, bool SPACING, bool WANT_E>
__device__ __forceinline__ void composite_forward_ray(
    const float* __restrict__ raw, int out_dim, const float* __restrict__ dens_in, const float* __restrict__ rgb_in,
    const float* __restrict__ target, int target_stride, float* __restrict__ dens_out, float* __restrict__ rgb_out,
    float* __restrict__ pixel, float* __restrict__ err, int S, SpacingArgs sp, int lane, int ray, float (&e)[3]) {
    // samples past the end of the ray re-read the last one (a valid address, no branch around the
    // load) and get alpha = 0, which takes them out of every sum and product below
    f32x4 y[SPL];
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = min(lane * SPL + i, S - 1);
        y[i] = fetch_sample<RAW>(raw, out_dim, dens_in, rgb_in, (size_t)ray * S + s);
    }
    float alpha[SPL], col[SPL][3];
    float local = 1.0f;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = lane * SPL + i;
        float d;
        activate_sample<RAW>(y[i], d, col[i]);
        if (RAW && dens_out && s < S) {
            const size_t p = (size_t)ray * S + s;
            dens_out[p] = d;
            rgb_out[p * 3 + 0] = col[i][0]; rgb_out[p * 3 + 1] = col[i][1]; rgb_out[p * 3 + 2] = col[i][2];
        }
        float delta = NERF_DELTA;  // nerf.c:35
        if (SPACING) {
            const float* off = sp.offsets ? sp.offsets + (size_t)ray * S : nullptr;
            delta = sample_spacing(min(s, S - 1), S, sp.near_plane, sp.far_plane, off);
        }
        const float a = 1.0f - expf_cr(-d * delta);
        alpha[i] = s < S ? a : 0.0f;
        local *= (1.0f - alpha[i]);
    }
    float T = wave_exclusive_product(local, lane);
    if (!SPACING) {
        // does any sample's transmittance sit too close to the stop for the scan to decide it?
        const float band = guard_band(S);
        bool near = false;
        float t = T;
#pragma unroll
        for (int i = 0; i < SPL; i++) {
            near = near || fabsf(t - NERF_T_STOP) <= band;
            t *= (1.0f - alpha[i]);
        }
        if (__builtin_amdgcn_ballot_w64(near) != 0ull) T = sequential_transmittance<SPL>(alpha, lane);  // wave-uniform
    }
    float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        if (SPACING || T >= NERF_T_STOP) {  // equivalent to nerf.c:42-43 because T is non-increasing
            const float w = alpha[i] * T;
            acc[0] += w * col[i][0]; acc[1] += w * col[i][1]; acc[2] += w * col[i][2];
        }
        T *= (1.0f - alpha[i]);
    }
    acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]); acc[2] = wave_sum(acc[2]);
    if (lane < 3) {
        const float a = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
        pixel[ray * 3 + lane] = a;
        if (err) err[ray * 3 + lane] = a - target[(size_t)ray * target_stride + lane];  // nerf.c:54-55
    }
    if (WANT_E) {
#pragma unroll
        for (int c = 0; c < 3; c++) e[c] = acc[c] - target[(size_t)ray * target_stride + c];
    }
}

template <int SPL, bool RAW, bool SPACING>
__global__ __launch_bounds__(256) void composite_forward_kernel(
    const float* __restrict__ raw, int out_dim, const float* __restrict__ dens_in, const float* __restrict__ rgb_in,
    const float* __restrict__ target, int target_stride, float* __restrict__ dens_out, float* __restrict__ rgb_out,
    float* __restrict__ pixel, float* __restrict__ err, int R, int S, SpacingArgs sp) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;  // whole wavefront leaves together
    float e[3];
    composite_forward_ray<SPL, RAW, SPACING, false>(raw, out_dim, dens_in, rgb_in, target, target_stride, dens_out, rgb_out,
                                                    pixel, err, S, sp, lane, ray, e);
}

// loss[0] += sum of squares of err[0..n): one workgroup, fixed summation order (reproducible;
// the reference's atomicAdd at nerf.c:59 is order-dependent).
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ err, int n, float* loss,
                                                           int accumulate = 1) {
    __shared__ float part[1024];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) s += err[i] * err[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = accumulate ? loss[0] + part[0] : part[0];
}

// One ray's gradient on one wavefront, from the ray's pixel error (e0, e1, e2)
template <int SPL, bool RAW, bool SPACING>
__device__ __forceinline__ void composite_backward_ray(
    const float* __restrict__ raw, int out_dim, const float* __restrict__ dens_in, const float* __restrict__ rgb_in,
    const float e0, const float e1, const float e2, float* __restrict__ d_out, int S, SpacingArgs sp, int lane, int ray) {
    f32x4 y[SPL];  // as in the forward kernel: all loads first, clamped to the ray
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = min(lane * SPL + i, S - 1);
        y[i] = fetch_sample<RAW>(raw, out_dim, dens_in, rgb_in, (size_t)ray * S + s);
    }
    float alpha[SPL], col[SPL][3], dens[SPL], q[SPL], delta[SPL], ex[SPL];
    float local = 1.0f;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
        const int s = lane * SPL + i;
        activate_sample<RAW>(y[i], dens[i], col[i]);
        delta[i] = NERF_DELTA;
        if (SPACING) {
            const float* off = sp.offsets ? sp.offsets + (size_t)ray * S : nullptr;
            delta[i] = sample_spacing(min(s, S - 1), S, sp.near_plane, sp.far_plane, off);
        }
        ex[i] = expf_cr(-dens[i] * delta[i]);                                   // nerf.c:79 and nerf.c:101 take the same value
        alpha[i] = s < S ? 1.0f - ex[i] : 0.0f;
        q[i] = e0 * col[i][0] + e1 * col[i][1] + e2 * col[i][2];                // nerf.c:105-107
        local *= (1.0f - alpha[i]);
    }
    // transmittance in front of each sample, no early stop (nerf.c:82-86)
    float Tbefore[SPL];
    {
        float T = wave_exclusive_product(local, lane);
#pragma unroll
        for (int i = 0; i < SPL; i++) {
            Tbefore[i] = T;
            T *= (1.0f - alpha[i]);
        }
    }
    if (SPACING) {
        // analytic gradient: tail_s = sum_{t>s} w_t q_t as a reverse additive scan
        float wq[SPL], lane_sum = 0.0f;
#pragma unroll
        for (int i = 0; i < SPL; i++) {
            wq[i] = alpha[i] * Tbefore[i] * q[i];
            lane_sum += wq[i];
        }
        float tail = wave_exclusive_suffix_sum(lane_sum, lane);  // everything in the lanes after mine
#pragma unroll
        for (int i = SPL - 1; i >= 0; i--) {
            const int s = lane * SPL + i;
            if (s < S) {
                const float w = alpha[i] * Tbefore[i];
                const float gd = delta[i] * ((1.0f - alpha[i]) * Tbefore[i] * q[i] - tail);
                f32x4 g;
                g[0] = dens[i] > 0.0f ? gd : 0.0f;
                g[1] = (w * e0) * (col[i][0] * (1.0f - col[i][0]));
                g[2] = (w * e1) * (col[i][1] * (1.0f - col[i][1]));
                g[3] = (w * e2) * (col[i][2] * (1.0f - col[i][2]));
                *reinterpret_cast<f32x4*>(d_out + ((size_t)ray * S + s) * 4) = g;
            }
            tail += wq[i];
        }
        return;
    }
    // alpha*q of the sample that follows each of mine (0 past the end of the ray)
    float aq_next[SPL];
    {
        const float first = alpha[0] * q[0];
        const float from_next_lane = __shfl_down(first, 1, 64);
#pragma unroll
        for (int i = 0; i < SPL - 1; i++) aq_next[i] = alpha[i + 1] * q[i + 1];
        aq_next[SPL - 1] = lane == 63 ? 0.0f : from_next_lane;
    }
    // this lane's composed map x -> A x + B over its samples, last sample applied first
    float A = 1.0f, B = 0.0f;
#pragma unroll
    for (int i = SPL - 1; i >= 0; i--) {
        const float a = 1.0f - alpha[i];
        const float b = a * aq_next[i];
        B = a * B + b;
        A = a * A;
    }
    // inclusive suffix composition across lanes: G_l = F_l o F_{l+1} o ... o F_63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float A2 = __shfl_down(A, d, 64);
        const float B2 = __shfl_down(B, d, 64);
        if (lane + d < 64) {
            B = A * B2 + B;
            A = A * A2;
        }
    }
    // R entering my last sample = G_{l+1}(0)
    float Rn = __shfl_down(B, 1, 64);
    if (lane == 63) Rn = 0.0f;

#pragma unroll
    for (int i = SPL - 1; i >= 0; i--) {
        const int s = lane * SPL + i;
        const float a = 1.0f - alpha[i];
        const float Rs = a * (aq_next[i] + Rn);
        if (s < S) {
            const float w = alpha[i] * Tbefore[i];                              // nerf.c:86
            const float da = NERF_DELTA * ex[i];                                // nerf.c:101
            const float gd = da * (Tbefore[i] * q[i] - Rs);
            f32x4 g;
            g[0] = dens[i] > 0.0f ? gd : 0.0f;                                  // nerf.c:123-124
            g[1] = (w * e0) * (col[i][0] * (1.0f - col[i][0]));                 // nerf.c:93-96
            g[2] = (w * e1) * (col[i][1] * (1.0f - col[i][1]));
            g[3] = (w * e2) * (col[i][2] * (1.0f - col[i][2]));
            *reinterpret_cast<f32x4*>(d_out + ((size_t)ray * S + s) * 4) = g;
        }
        Rn = Rs;
    }
}

template <int SPL, bool RAW, bool SPACING>
__global__ __launch_bounds__(256) void composite_backward_kernel(
    const float* __restrict__ raw, int out_dim, const float* __restrict__ dens_in, const float* __restrict__ rgb_in,
    const float* __restrict__ err, float* __restrict__ d_out, int R, int S, SpacingArgs sp) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    composite_backward_ray<SPL, RAW, SPACING>(raw, out_dim, dens_in, rgb_in, err[ray * 3 + 0], err[ray * 3 + 1],
                                              err[ray * 3 + 2], d_out, S, sp, lane, ray);
}

// Forward pass and gradient of a ray in one launch, for the training step (the gradient of a ray needs that
// ray's pixel error and nothing of any other ray): the two routines above one after the other on the same
// wavefront, so every value is the one the two kernels compute -- the samples, their activations and
// exponentials are simply still in registers -- and a launch and a second read of the MLP's output are saved.
template <int SPL, bool SPACING>
__global__ __launch_bounds__(256) void composite_forward_backward_kernel(
    const float* __restrict__ raw, int out_dim, const float* __restrict__ target, int target_stride,
    float* __restrict__ pixel, float* __restrict__ err, float* __restrict__ d_out, int R, int S, SpacingArgs sp) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    float e[3];
    composite_forward_ray<SPL, true, SPACING, true>(raw, out_dim, nullptr, nullptr, target, target_stride, nullptr, nullptr,
                                                    pixel, err, S, sp, lane, ray, e);
    composite_backward_ray<SPL, true, SPACING>(raw, out_dim, nullptr, nullptr, e[0], e[1], e[2], d_out, S, sp, lane, ray);
}

__global__ __launch_bounds__(256) void activation_kernel(const float* __restrict__ y, float* __restrict__ dens,
                                                         float* __restrict__ rgb, int n, int out_dim) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* o = y + (size_t)i * out_dim;
    dens[i] = fmaxf(0.0f, o[0]);
    rgb[i * 3 + 0] = sigmoid_ref(o[1]);  // the arithmetic of activate_sample<true>
    rgb[i * 3 + 1] = sigmoid_ref(o[2]);
    rgb[i * 3 + 2] = sigmoid_ref(o[3]);
}

// offsets[i] = uniform [0, 1) from a counter-based generator (splitmix64 of seed, batch and index):
// one jittered sample per spacing interval, reproducible and independent of launch geometry
__global__ __launch_bounds__(256) void stratified_offsets_kernel(float* __restrict__ out, long count,
                                                                 unsigned long long seed, unsigned long long batch) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (batch * 0x100000001B3ULL + (unsigned long long)i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    out[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
}

int spl_for(int S) {
    const int need = (S + 63) / 64;
    for (int spl = 1; spl <= 16; spl <<= 1)
        if (need <= spl) return spl;
    return 0;
}

template <bool RAW, bool SPACING = false>
int launch_forward(const float* raw, int out_dim, const float* dens_in, const float* rgb_in, const float* target,
                   int target_stride, float* dens_out, float* rgb_out, float* pixel, float* err, int R, int S,
                   SpacingArgs sp = SpacingArgs{nullptr, 0.0f, 0.0f}) {
    const dim3 grid((R + 3) / 4), block(256);
#define FWD(SPL)                                                                                                 \
    hipLaunchKernelGGL((composite_forward_kernel<SPL, RAW, SPACING>), grid, block, 0, stream(), raw, out_dim,    \
                       dens_in, rgb_in, target, target_stride, dens_out, rgb_out, pixel, err, R, S, sp)
    switch (spl_for(S)) {
        case 1: FWD(1); break;
        case 2: FWD(2); break;
        case 4: FWD(4); break;
        case 8: FWD(8); break;
        case 16: FWD(16); break;
        default: return NERF_ERR_ARG;
    }
#undef FWD
    return NERF_OK;
}

template <bool RAW, bool SPACING = false>
int launch_backward(const float* raw, int out_dim, const float* dens_in, const float* rgb_in, const float* err,
                    float* d_out, int R, int S, SpacingArgs sp = SpacingArgs{nullptr, 0.0f, 0.0f}) {
    const dim3 grid((R + 3) / 4), block(256);
#define BWD(SPL)                                                                                                 \
    hipLaunchKernelGGL((composite_backward_kernel<SPL, RAW, SPACING>), grid, block, 0, stream(), raw, out_dim,   \
                       dens_in, rgb_in, err, d_out, R, S, sp)
    switch (spl_for(S)) {
        case 1: BWD(1); break;
        case 2: BWD(2); break;
        case 4: BWD(4); break;
        case 8: BWD(8); break;
        case 16: BWD(16); break;
        default: return NERF_ERR_ARG;
    }
#undef BWD
    return NERF_OK;
}

template <bool SPACING>
int launch_forward_backward(const float* raw, int out_dim, const float* target, int target_stride, float* pixel, float* err,
                            float* d_out, int R, int S, SpacingArgs sp = SpacingArgs{nullptr, 0.0f, 0.0f}) {
    const dim3 grid((R + 3) / 4), block(256);
#define FB(SPL)                                                                                                  \
    hipLaunchKernelGGL((composite_forward_backward_kernel<SPL, SPACING>), grid, block, 0, stream(), raw, out_dim, \
                       target, target_stride, pixel, err, d_out, R, S, sp)
    switch (spl_for(S)) {
        case 1: FB(1); break;
        case 2: FB(2); break;
        case 4: FB(4); break;
        case 8: FB(8); break;
        case 16: FB(16); break;
        default: return NERF_ERR_ARG;
    }
#undef FB
    return NERF_OK;
}

}  // namespace

// loss[0] = (accumulate ? loss[0] : 0) + sum of squares of err[R][3] (nerf.c:52-59), fixed summation order
int loss_sum(const float* d_err, int R, float* d_loss, int accumulate) {
    if (R <= 0 || !d_err || !d_loss) return NERF_ERR_ARG;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, stream(), d_err, R * 3, d_loss, accumulate);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}
}  // namespace nerf

using namespace nerf;

extern "C" {

// ---- the reference's three kernels as C launchers (declared in include/nerf_kernels.h) ----------

void activation_kernel_launch(float* d_layer_output, float* d_densities, float* d_colors, int batch_size,
                              int output_dim) {
    if (batch_size <= 0) return;
    hipLaunchKernelGGL(activation_kernel, dim3((batch_size + 255) / 256), dim3(256), 0, stream(), d_layer_output,
                       d_densities, d_colors, batch_size, output_dim);
    NERF_HIP_CHECK(hipGetLastError());
}

void volume_rendering_and_loss_launch(const float* d_densities, const float* d_colors, const float* d_true_colors,
                                      float* d_pixel_colors, float* d_pixel_errors, float* d_loss_accum,
                                      int rays_per_batch, int num_samples) {
    if (rays_per_batch <= 0) return;
    if (launch_forward<false>(nullptr, 0, d_densities, d_colors, d_true_colors, num_samples * 3, nullptr, nullptr,
                              d_pixel_colors, d_pixel_errors, rays_per_batch, num_samples) != NERF_OK) {
        fprintf(stderr, "volume_rendering_and_loss_launch: num_samples %d not supported (max 1024)\n", num_samples);
        exit(EXIT_FAILURE);
    }
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, stream(), d_pixel_errors, rays_per_batch * 3,
                       d_loss_accum, 1);
    NERF_HIP_CHECK(hipGetLastError());
}

void volume_rendering_gradient_launch(const float* d_densities, const float* d_colors, const float* d_pixel_errors,
                                      float* d_mlp_error_output, int rays_per_batch, int num_samples) {
    if (rays_per_batch <= 0) return;
    if (launch_backward<false>(nullptr, 0, d_densities, d_colors, d_pixel_errors, d_mlp_error_output, rays_per_batch,
                               num_samples) != NERF_OK) {
        fprintf(stderr, "volume_rendering_gradient_launch: num_samples %d not supported (max 1024)\n", num_samples);
        exit(EXIT_FAILURE);
    }
    NERF_HIP_CHECK(hipGetLastError());
}

// ---- fused forms used by the trainer and the renderer (declared in include/nerf_amd.h) -----------

int nerf_composite_forward(const float* d_raw, int out_dim, const float* d_target, int target_stride, float* d_dens,
                           float* d_rgb, float* d_pixel, float* d_err, int R, int S) {
    if (R <= 0 || S <= 0 || out_dim < 4 || !d_raw || !d_pixel) return NERF_ERR_ARG;
    if (d_err && !d_target) return NERF_ERR_ARG;
    if ((d_dens == nullptr) != (d_rgb == nullptr)) return NERF_ERR_ARG;
    const int rc = launch_forward<true>(d_raw, out_dim, nullptr, nullptr, d_target, target_stride, d_dens, d_rgb,
                                        d_pixel, d_err, R, S);
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

// ---- opt-in quadrature with the true sample spacing (no counterpart in the reference) --------------

int nerf_composite_forward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                   float far_plane, const float* d_target, int target_stride, float* d_pixel,
                                   float* d_err, int R, int S) {
    if (R <= 0 || S <= 1 || out_dim < 4 || !d_raw || !d_pixel) return NERF_ERR_ARG;
    if (d_err && !d_target) return NERF_ERR_ARG;
    const int rc = launch_forward<true, true>(d_raw, out_dim, nullptr, nullptr, d_target, target_stride, nullptr, nullptr,
                                              d_pixel, d_err, R, S, SpacingArgs{d_offsets, near_plane, far_plane});
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_composite_backward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                    float far_plane, const float* d_err, float* d_error_output, int R, int S) {
    if (R <= 0 || S <= 1 || out_dim != 4 || !d_raw || !d_err || !d_error_output) return NERF_ERR_ARG;
    const int rc = launch_backward<true, true>(d_raw, out_dim, nullptr, nullptr, d_err, d_error_output, R, S,
                                               SpacingArgs{d_offsets, near_plane, far_plane});
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_stratified_offsets(float* d_offsets, long count, unsigned long long seed, unsigned long long batch) {
    if (!d_offsets || count <= 0) return NERF_ERR_ARG;
    hipLaunchKernelGGL(stratified_offsets_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream(),
                       d_offsets, count, seed, batch);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_composite_forward_backward(const float* d_raw, int out_dim, const float* d_target, int target_stride,
                                    float* d_pixel, float* d_err, float* d_error_output, int R, int S) {
    if (R <= 0 || S <= 0 || out_dim != 4 || !d_raw || !d_target || !d_pixel || !d_err || !d_error_output) return NERF_ERR_ARG;
    const int rc = launch_forward_backward<false>(d_raw, out_dim, d_target, target_stride, d_pixel, d_err, d_error_output, R, S);
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_composite_forward_backward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                            float far_plane, const float* d_target, int target_stride, float* d_pixel,
                                            float* d_err, float* d_error_output, int R, int S) {
    if (R <= 0 || S <= 1 || out_dim != 4 || !d_raw || !d_target || !d_pixel || !d_err || !d_error_output) return NERF_ERR_ARG;
    const int rc = launch_forward_backward<true>(d_raw, out_dim, d_target, target_stride, d_pixel, d_err, d_error_output, R, S,
                                                 SpacingArgs{d_offsets, near_plane, far_plane});
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_loss_accumulate(const float* d_err, int R, float* d_loss) { return nerf::loss_sum(d_err, R, d_loss, 1); }

int nerf_composite_backward(const float* d_raw, int out_dim, const float* d_err, float* d_error_output, int R, int S) {
    if (R <= 0 || S <= 0 || out_dim != 4 || !d_raw || !d_err || !d_error_output) return NERF_ERR_ARG;
    const int rc = launch_backward<true>(d_raw, out_dim, nullptr, nullptr, d_err, d_error_output, R, S);
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

}  // extern "C"
