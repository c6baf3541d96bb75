// Per-Gaussian projection ("preprocess") forward and backward for gfx950.
//
// One thread owns one Gaussian of one scene and walks that scene's V views, so the Gaussian's
// parameters (mean, scale, quaternion, SH block) are read from HBM once per scene instead of once
// per (scene, view) -- the reference materialises v copies of every Gaussian tensor before calling
// its rasterizer (/root/reference/src/model/decoder/decoder_splatting_cuda.py:59-64).
//
// Semantics: SURVEY.md Appendix B #1-#9 (restated in oracle/splat_ref.py::project).
#include <atomic>

#include "spf_common.h"

namespace spf {

// Camera tables (view / projection matrix, tan fov, scale of a render) are the same for every thread of a block and
// were written by an EARLIER launch: read through the constant address space they become scalar loads into SGPRs.
// Through an ordinary pointer the compiler cannot rule out that the kernel's own stores alias them and falls back to
// per-lane vector loads -- 44 floats and 12 doubles per view in VGPRs (the float64 pose alone cost one wave per SIMD).
typedef const float __attribute__((address_space(4))) * kfloat_p;
typedef const double __attribute__((address_space(4))) * kdouble_p;
__device__ __forceinline__ kfloat_p as_const(const float* p) { return (kfloat_p)(uintptr_t)p; }
__device__ __forceinline__ kdouble_p as_const(const double* p) { return (kdouble_p)(uintptr_t)p; }

__device__ __forceinline__ void quat_rot(const float4 q, float R[9]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// N = R diag(s): the 3-D covariance is N N^T and is never formed.  With M = J W (2x3) the 2-D
// covariance is B B^T + 0.3 I, B = M N: its entries are dot products of the rows of B and its determinant is the sum of
// the squared 2x2 minors of B plus low-pass terms (see stable_det) -- shorter rounding chains than M (N N^T) M^T, nine
// live registers per Gaussian, and the backward accumulates dL/dN directly.
__device__ __forceinline__ void scale_columns(const float R[9], float sx, float sy, float sz, float N[9]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { N[3 * j] = R[3 * j] * sx; N[3 * j + 1] = R[3 * j + 1] * sy; N[3 * j + 2] = R[3 * j + 2] * sz; }
}

// Everything the forward and the backward need about one (view, Gaussian) projection.
struct Proj {
    float tx, ty, tz;          // view-space position
    float homx, homy, pw;      // clip x, y and 1/(w+eps)
    float px, py;              // pixel centre
    float tcx, tcy;            // frustum-clamped t.x, t.y used in J
    bool inx, iny;             // clamp inactive
    float m0[3], m1[3];        // rows of M = J * Wcv
    float J00, J02, J11, J12;
    float fx, fy, itz;         // focal lengths in pixels, 1 / t.z
    float b0[3], b1[3];        // rows of B = M N (N already carries the render's world scale)
    float a, b, c;             // 2-D covariance B B^T + 0.3 I
};

// `p` = the mean after the per-render world scale, `p0` = before it; `M64` (may be null) = the float64 view matrix with
// that scale folded in (SpfInputs.viewmatrix64).  The view-space position is formed in float64 and rounded once: it is
// a difference of terms as large as the camera translation, and after the reference's 1/near rescale a Gaussian at
// z = 0.25 sits behind a translation of tens of units -- float32 lost five digits of z there (and 1/z^2 scales the
// whole footprint).  12 half-rate fmas per (Gaussian, view).
__device__ __forceinline__ void project_point(const float p[3], const float p0[3], kfloat_p Vm, kdouble_p M64,
                                              kfloat_p Pm, float tanx, float tany, float fx, float fy,
                                              int H, int W, const float N[9], float nscale, Proj& o) {
    if (M64) {
        const double x = p0[0], y = p0[1], z = p0[2];
        o.tx = (float)__builtin_fma(x, M64[0], __builtin_fma(y, M64[4], __builtin_fma(z, M64[8], M64[12])));
        o.ty = (float)__builtin_fma(x, M64[1], __builtin_fma(y, M64[5], __builtin_fma(z, M64[9], M64[13])));
        o.tz = (float)__builtin_fma(x, M64[2], __builtin_fma(y, M64[6], __builtin_fma(z, M64[10], M64[14])));
    } else {
        const double x = p[0], y = p[1], z = p[2];
        o.tx = (float)__builtin_fma(x, (double)Vm[0], __builtin_fma(y, (double)Vm[4], __builtin_fma(z, (double)Vm[8], (double)Vm[12])));
        o.ty = (float)__builtin_fma(x, (double)Vm[1], __builtin_fma(y, (double)Vm[5], __builtin_fma(z, (double)Vm[9], (double)Vm[13])));
        o.tz = (float)__builtin_fma(x, (double)Vm[2], __builtin_fma(y, (double)Vm[6], __builtin_fma(z, (double)Vm[10], (double)Vm[14])));
    }
    o.homx = o.tx * Pm[0] + o.ty * Pm[4] + o.tz * Pm[8] + Pm[12];
    o.homy = o.tx * Pm[1] + o.ty * Pm[5] + o.tz * Pm[9] + Pm[13];
    const float homw = o.tx * Pm[3] + o.ty * Pm[7] + o.tz * Pm[11] + Pm[15];
    o.pw = 1.0f / (homw + 1e-7f);
    o.px = ((o.homx * o.pw + 1.0f) * W - 1.0f) * 0.5f;
    o.py = ((o.homy * o.pw + 1.0f) * H - 1.0f) * 0.5f;

    // `fx`, `fy` = W / (2 tan fov_x), H / (2 tan fov_y): the same for every Gaussian of a view, so the two IEEE divisions
    // are done once per block and view (view_focal), not once per (Gaussian, view).
    o.fx = fx; o.fy = fy;
    // Frustum clamp: t.x / t.z is held inside +-1.3 tan fov.  |t.x / t.z| <= lim is decided as
    // |t.x| <= lim * t.z (t.z > 0 for every Gaussian that is kept) and a clamped t.x is +-lim * t.z outright: two IEEE
    // divisions per (Gaussian, view) less.  Value and gradient are continuous across the boundary, so the one-ulp
    // difference between the two forms of the test moves nothing.
    const float limx = kFovClamp * tanx * o.tz, limy = kFovClamp * tany * o.tz;
    o.inx = fabsf(o.tx) <= limx;
    o.iny = fabsf(o.ty) <= limy;
    o.tcx = o.inx ? o.tx : copysignf(limx, o.tx);
    o.tcy = o.iny ? o.ty : copysignf(limy, o.ty);
    const float itz = 1.0f / o.tz;
    o.itz = itz;
    o.J00 = fx * itz;
    o.J11 = fy * itz;
    o.J02 = -fx * o.tcx * itz * itz;
    o.J12 = -fy * o.tcy * itz * itz;
    // Wcv[i][j] = Vm[4j + i]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o.m0[j] = o.J00 * Vm[4 * j + 0] + o.J02 * Vm[4 * j + 2];
        o.m1[j] = o.J11 * Vm[4 * j + 1] + o.J12 * Vm[4 * j + 2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.b0[k] = nscale * (o.m0[0] * N[k] + o.m0[1] * N[3 + k] + o.m0[2] * N[6 + k]);
        o.b1[k] = nscale * (o.m1[0] * N[k] + o.m1[1] * N[3 + k] + o.m1[2] * N[6 + k]);
    }
    o.a = o.b0[0] * o.b0[0] + o.b0[1] * o.b0[1] + o.b0[2] * o.b0[2] + kLowPass;
    o.b = o.b0[0] * o.b1[0] + o.b0[1] * o.b1[1] + o.b0[2] * o.b1[2];
    o.c = o.b1[0] * o.b1[0] + o.b1[1] * o.b1[1] + o.b1[2] * o.b1[2] + kLowPass;
}

// Focal lengths of the scene's views, one view per LANE (lane v holds view v's pair; V <= 64): the view loop then
// fetches them with v_readlane (the view index is uniform).  Scenes with more views fall back to the division per view.
struct LaneFocal { float fx, fy; };
__device__ __forceinline__ LaneFocal lane_focal(const float* __restrict__ tanfov, int s, int V, int H, int W, int lane) {
    LaneFocal f = {0.f, 0.f};
    if (V <= kWave && lane < V) {
        const float tanx = tanfov[2 * (s * V + lane)], tany = tanfov[2 * (s * V + lane) + 1];
        f.fx = W / (2.0f * tanx);
        f.fy = H / (2.0f * tany);
    }
    // The values are read ACROSS lanes later (v_readlane).  The compiler does not know that: left free, it sinks the two
    // divisions into the divergent region that holds their only "use" (the backward's `if (vis)`), where the lane that
    // owns a view's pair may be inactive -- degree 4 of the backward did exactly that.  Opaque here, in uniform flow.
    asm volatile("" : "+v"(f.fx), "+v"(f.fy));
    return f;
}
// LANES = false (degree >= 2: the kernels are bound by the coefficient traffic and short of registers, measured 2 - 4 %
// slower with the per-lane pair and its early wait): the plain division per view.
template <bool LANES>
__device__ __forceinline__ void view_focal(const LaneFocal& f, int V, int v, float tanx, float tany, int H, int W,
                                           float& fx, float& fy) {
    if (LANES && V <= kWave) {
        fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.fx), v));
        fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.fy), v));
    } else {
        fx = W / (2.0f * tanx);
        fy = H / (2.0f * tany);
    }
}

// det of the low-passed 2-D covariance WITHOUT the cancellation of a*c - b*b.  By Cauchy-Binet det(B B^T) is the sum
// of the squared 2x2 minors of B (2x3), hence
//     det = sum_{i<j} minor_ij(B)^2 + 0.3 (a0 + c0) + 0.09        (a0, c0: diagonal before the low-pass)
// -- every term is >= 0.  For a thin splat hundreds of pixels long a*c and b*b are ~1e12 while their difference is
// ~1e6: float32 loses the leading digits of det (any float32 evaluation of the classic expression does), and 1/det
// scales the whole exponent of every pixel the splat touches (fuzz seeds 2135 / 2195 / 2389 of round 2's wide family:
// colours off by up to 2.4e-3).  This form is good to a few ulp whatever the aspect ratio.
__device__ __forceinline__ float diff_of_products(float a, float b, float c, float d) {   // a*b - c*d, compensated
    const float w = c * d;
    const float e = fmaf(-c, d, w);
    return fmaf(a, b, -w) + e;
}
__device__ __forceinline__ float stable_det(const Proj& o) {
    const float n01 = diff_of_products(o.b0[0], o.b1[1], o.b0[1], o.b1[0]);
    const float n02 = diff_of_products(o.b0[0], o.b1[2], o.b0[2], o.b1[0]);
    const float n12 = diff_of_products(o.b0[1], o.b1[2], o.b0[2], o.b1[1]);
    const float tr0 = (o.a - kLowPass) + (o.c - kLowPass);
    return n01 * n01 + n02 * n02 + n12 * n12 + kLowPass * tr0 + kLowPass * kLowPass;
}

// Real-SH basis function k (0..24; index n(n+1)+m, the 3DGS family's signs) at the unit direction d, optionally with
// the partial derivatives of the polynomial as written (the caller projects them onto the tangent space of the unit
// sphere).  `k` is a compile-time constant at every call site (fully unrolled loops), so the switch folds away; terms
// are evaluated one at a time, right where they are consumed, instead of into 4 x 25-entry arrays -- the projection
// backward with 16 / 25 coefficients was register-bound (300+ VGPRs, one wave per SIMD).
struct ShDir {
    float x, y, z, xx, yy, zz, xy, yz, xz;
};
__device__ __forceinline__ ShDir sh_dir(float x, float y, float z) {
    return ShDir{x, y, z, x * x, y * y, z * z, x * y, y * z, x * z};
}
template <bool WITH_GRAD>
__device__ __forceinline__ float sh_term(int k, const ShDir& d, float& gx, float& gy, float& gz) {
    const float x = d.x, y = d.y, z = d.z, xx = d.xx, yy = d.yy, zz = d.zz, xy = d.xy, yz = d.yz, xz = d.xz;
    float b = 0.f;
    gx = gy = gz = 0.f;
    switch (k) {
        case 0: b = SH_C0; break;
        case 1: b = -SH_C1 * y; if (WITH_GRAD) gy = -SH_C1; break;
        case 2: b = SH_C1 * z; if (WITH_GRAD) gz = SH_C1; break;
        case 3: b = -SH_C1 * x; if (WITH_GRAD) gx = -SH_C1; break;
        case 4: b = SH_C2[0] * xy; if (WITH_GRAD) { gx = SH_C2[0] * y; gy = SH_C2[0] * x; } break;
        case 5: b = SH_C2[1] * yz; if (WITH_GRAD) { gy = SH_C2[1] * z; gz = SH_C2[1] * y; } break;
        case 6: b = SH_C2[2] * (2.f * zz - xx - yy);
            if (WITH_GRAD) { gx = SH_C2[2] * -2.f * x; gy = SH_C2[2] * -2.f * y; gz = SH_C2[2] * 4.f * z; } break;
        case 7: b = SH_C2[3] * xz; if (WITH_GRAD) { gx = SH_C2[3] * z; gz = SH_C2[3] * x; } break;
        case 8: b = SH_C2[4] * (xx - yy); if (WITH_GRAD) { gx = SH_C2[4] * 2.f * x; gy = SH_C2[4] * -2.f * y; } break;
        case 9: b = SH_C3[0] * y * (3.f * xx - yy);
            if (WITH_GRAD) { gx = SH_C3[0] * 6.f * xy; gy = SH_C3[0] * (3.f * xx - 3.f * yy); } break;
        case 10: b = SH_C3[1] * xy * z; if (WITH_GRAD) { gx = SH_C3[1] * yz; gy = SH_C3[1] * xz; gz = SH_C3[1] * xy; } break;
        case 11: b = SH_C3[2] * y * (4.f * zz - xx - yy);
            if (WITH_GRAD) { gx = SH_C3[2] * -2.f * xy; gy = SH_C3[2] * (4.f * zz - xx - 3.f * yy); gz = SH_C3[2] * 8.f * yz; } break;
        case 12: b = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            if (WITH_GRAD) { gx = SH_C3[3] * -6.f * xz; gy = SH_C3[3] * -6.f * yz; gz = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy); } break;
        case 13: b = SH_C3[4] * x * (4.f * zz - xx - yy);
            if (WITH_GRAD) { gx = SH_C3[4] * (4.f * zz - 3.f * xx - yy); gy = SH_C3[4] * -2.f * xy; gz = SH_C3[4] * 8.f * xz; } break;
        case 14: b = SH_C3[5] * z * (xx - yy);
            if (WITH_GRAD) { gx = SH_C3[5] * 2.f * xz; gy = SH_C3[5] * -2.f * yz; gz = SH_C3[5] * (xx - yy); } break;
        case 15: b = SH_C3[6] * x * (xx - 3.f * yy);
            if (WITH_GRAD) { gx = SH_C3[6] * (3.f * xx - 3.f * yy); gy = SH_C3[6] * -6.f * xy; } break;
        case 16: b = SH_C4[0] * xy * (xx - yy);
            if (WITH_GRAD) { gx = SH_C4[0] * y * (3.f * xx - yy); gy = SH_C4[0] * x * (xx - 3.f * yy); } break;
        case 17: b = SH_C4[1] * yz * (3.f * xx - yy);
            if (WITH_GRAD) { gx = SH_C4[1] * 6.f * xy * z; gy = SH_C4[1] * z * (3.f * xx - 3.f * yy); gz = SH_C4[1] * y * (3.f * xx - yy); } break;
        case 18: b = SH_C4[2] * xy * (7.f * zz - 1.f);
            if (WITH_GRAD) { gx = SH_C4[2] * y * (7.f * zz - 1.f); gy = SH_C4[2] * x * (7.f * zz - 1.f); gz = SH_C4[2] * 14.f * xy * z; } break;
        case 19: b = SH_C4[3] * yz * (7.f * zz - 3.f);
            if (WITH_GRAD) { gy = SH_C4[3] * z * (7.f * zz - 3.f); gz = SH_C4[3] * y * (21.f * zz - 3.f); } break;
        case 20: b = SH_C4[4] * (zz * (35.f * zz - 30.f) + 3.f); if (WITH_GRAD) gz = SH_C4[4] * z * (140.f * zz - 60.f); break;
        case 21: b = SH_C4[5] * xz * (7.f * zz - 3.f);
            if (WITH_GRAD) { gx = SH_C4[5] * z * (7.f * zz - 3.f); gz = SH_C4[5] * x * (21.f * zz - 3.f); } break;
        case 22: b = SH_C4[6] * (xx - yy) * (7.f * zz - 1.f);
            if (WITH_GRAD) { gx = SH_C4[6] * 2.f * x * (7.f * zz - 1.f); gy = SH_C4[6] * -2.f * y * (7.f * zz - 1.f); gz = SH_C4[6] * 14.f * z * (xx - yy); } break;
        case 23: b = SH_C4[7] * xz * (xx - 3.f * yy);
            if (WITH_GRAD) { gx = SH_C4[7] * z * (3.f * xx - 3.f * yy); gy = SH_C4[7] * -6.f * xy * z; gz = SH_C4[7] * x * (xx - 3.f * yy); } break;
        case 24: b = SH_C4[8] * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
            if (WITH_GRAD) { gx = SH_C4[8] * 4.f * x * (xx - 3.f * yy); gy = SH_C4[8] * 4.f * y * (yy - 3.f * xx); } break;
        default: break;
    }
    return b;
}
// which partial derivatives of term k are structurally zero (so that no `0 * coefficient` is ever issued)
__device__ __forceinline__ constexpr bool sh_has_gx(int k) { return k != 0 && k != 1 && k != 2 && k != 5 && k != 19 && k != 20; }
__device__ __forceinline__ constexpr bool sh_has_gy(int k) { return k != 0 && k != 2 && k != 3 && k != 7 && k != 20 && k != 21; }
__device__ __forceinline__ constexpr bool sh_has_gz(int k) { return k != 0 && k != 1 && k != 3 && k != 4 && k != 8 && k != 9 && k != 15 && k != 16 && k != 24; }

// SH coefficient block of one Gaussian: K coefficients per channel, stored [K,3] (layout 0) or [3,K] (layout 1).
// Coefficients are consumed four at a time through three 16-byte accesses (any layout), the remaining NB % 4 one by
// one.  ALIGNED (K % 4 == 0): the accesses are 16-byte aligned; otherwise (K = 25, the reference's default d_sh) the
// block of a Gaussian starts at a multiple of 4 bytes only and the same dwordx4 instructions are issued with dword
// alignment (legal for global memory on gfx9+; a 75-float block read through 4-byte loads costs 4x the instructions
// and address-coalescer cycles).  v[i][c] = coefficient 4*k4+i of channel c.
template <bool ALIGNED>
__device__ __forceinline__ f4a ld4(const float* __restrict__ p) {
    if (ALIGNED) return *reinterpret_cast<const f4a*>(p);
    return *reinterpret_cast<const f4u*>(p);
}
template <bool ALIGNED>
__device__ __forceinline__ void st4(float* __restrict__ p, float a, float b, float c, float d) {
    const f4a v = {a, b, c, d};
    if (ALIGNED) *reinterpret_cast<f4a*>(p) = v;
    else *reinterpret_cast<f4u*>(p) = v;
}
// NATIVE == 2 (band split, SpfDims.sh_layout 2): coefficients 0..15 of channel c at sh[16 c + k] (16-byte aligned rows),
// 16..24 at hi[9 c + k - 16] (dword aligned).  Only the degree-4 kernels are instantiated with it: up to degree 3 the
// launchers hand plane 0 to the NATIVE == 1 kernels as a K = 16 block (launch_project_fwd / _bwd).
// NATIVE == 3 (raw rows, SpfDims.sh_layout 3): `sh` points at channel 7 of the Gaussian's RAW row -- the native [3][K]
// arrangement at dword alignment -- and every coefficient is scaled by the adapter's sh_mask[k] as it is read (`mk`, in the
// constant address space: scalar loads): raw * mask rounded to float32 first, exactly the value spf_adapter_forward
// would have stored (the two-pass path's numbers, to the rounding of a differently contracted multiply-add).
template <int NATIVE, bool ALIGNED>
__device__ __forceinline__ void sh_load4(const float* __restrict__ sh, const float* __restrict__ hi, int K, int k4,
                                         float v[4][3], kfloat_p mk = nullptr) {
    if (NATIVE == 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f4a t = ld4<false>(sh + c * K + 4 * k4);
            v[0][c] = t.x * mk[4 * k4]; v[1][c] = t.y * mk[4 * k4 + 1]; v[2][c] = t.z * mk[4 * k4 + 2]; v[3][c] = t.w * mk[4 * k4 + 3];
        }
    } else if (NATIVE == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f4a t = k4 < 4 ? ld4<true>(sh + c * 16 + 4 * k4) : ld4<false>(hi + c * 9 + 4 * (k4 - 4));
            v[0][c] = t.x; v[1][c] = t.y; v[2][c] = t.z; v[3][c] = t.w;
        }
    } else if (NATIVE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f4a t = ld4<ALIGNED>(sh + c * K + 4 * k4);
            v[0][c] = t.x; v[1][c] = t.y; v[2][c] = t.z; v[3][c] = t.w;
        }
    } else {
        const f4a a = ld4<ALIGNED>(sh + 12 * k4), b = ld4<ALIGNED>(sh + 12 * k4 + 4), c = ld4<ALIGNED>(sh + 12 * k4 + 8);
        v[0][0] = a.x; v[0][1] = a.y; v[0][2] = a.z; v[1][0] = a.w;
        v[1][1] = b.x; v[1][2] = b.y; v[2][0] = b.z; v[2][1] = b.w;
        v[2][2] = c.x; v[3][0] = c.y; v[3][1] = c.z; v[3][2] = c.w;
    }
}
template <int NATIVE, bool ALIGNED>
__device__ __forceinline__ void sh_store4(float* __restrict__ sh, float* __restrict__ hi, int K, int k4, const float v[4][3]) {
    if (NATIVE == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (k4 < 4) st4<true>(sh + c * 16 + 4 * k4, v[0][c], v[1][c], v[2][c], v[3][c]);
            else st4<false>(hi + c * 9 + 4 * (k4 - 4), v[0][c], v[1][c], v[2][c], v[3][c]);
        }
    } else if (NATIVE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) st4<ALIGNED>(sh + c * K + 4 * k4, v[0][c], v[1][c], v[2][c], v[3][c]);
    } else {
        st4<ALIGNED>(sh + 12 * k4, v[0][0], v[0][1], v[0][2], v[1][0]);
        st4<ALIGNED>(sh + 12 * k4 + 4, v[1][1], v[1][2], v[2][0], v[2][1]);
        st4<ALIGNED>(sh + 12 * k4 + 8, v[2][2], v[3][0], v[3][1], v[3][2]);
    }
}
// zero n floats at p (dword aligned): the gradient of coefficients that are carried but not evaluated
__device__ __forceinline__ void zero_floats(float* __restrict__ p, int n) {
    int i = 0;
    for (; i + 4 <= n; i += 4) st4<false>(p + i, 0.f, 0.f, 0.f, 0.f);
    for (; i < n; ++i) p[i] = 0.f;
}
template <int NATIVE>
__device__ __forceinline__ float sh_at(const float* __restrict__ sh, const float* __restrict__ hi, int K, int k, int c,
                                       kfloat_p mk = nullptr) {
    if (NATIVE == 3) return sh[c * K + k] * mk[k];
    if (NATIVE == 2) return k < 16 ? sh[c * 16 + k] : hi[c * 9 + k - 16];
    return NATIVE ? sh[c * K + k] : sh[3 * k + c];
}
// where coefficient k of channel c of a gradient block goes
template <int NATIVE>
__device__ __forceinline__ float* sh_slot(float* __restrict__ o, float* __restrict__ o_hi, int K, int k, int c) {
    if (NATIVE == 2) return k < 16 ? o + c * 16 + k : o_hi + c * 9 + k - 16;
    return NATIVE ? o + c * K + k : o + 3 * k + c;
}

// colour = sum_k basis_k sh_k (forward), optionally with D{x,y,z}[c] = sum_k dbasis_k/d{x,y,z} sh_k[c] (backward).
// The forward and the backward both come through here, so they take the same clamp decisions on the colour.
template <bool WITH_GRAD>
__device__ __forceinline__ void sh_accumulate(int k, const ShDir& dir, const float v[3], float col[3], float Dx[3],
                                              float Dy[3], float Dz[3]) {
    float gx, gy, gz;
    const float b = sh_term<WITH_GRAD>(k, dir, gx, gy, gz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        col[c] += b * v[c];
        if (WITH_GRAD) {
            if (sh_has_gx(k)) Dx[c] += gx * v[c];
            if (sh_has_gy(k)) Dy[c] += gy * v[c];
            if (sh_has_gz(k)) Dz[c] += gz * v[c];
        }
    }
}
template <int NB, int NATIVE, bool ALIGNED, bool WITH_GRAD>
__device__ __forceinline__ void sh_contract(const float* __restrict__ sh, const float* __restrict__ hi, int K,
                                            const ShDir& dir, float col[3], float Dx[3], float Dy[3], float Dz[3],
                                            kfloat_p mk = nullptr) {
    constexpr int NV = NB / 4;
    if (NV <= 1) {
#pragma unroll
        for (int k4 = 0; k4 < NV; ++k4) {
            float v[4][3];
            sh_load4<NATIVE, ALIGNED>(sh, hi, K, k4, v, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) sh_accumulate<WITH_GRAD>(4 * k4 + i, dir, v[i], col, Dx, Dy, Dz);
        }
    } else {
        // Groups of four coefficients, double-buffered by hand: group k4+1 is in flight while group k4 is consumed, and
        // a compiler barrier after every group keeps the scheduler from hoisting ALL the loads to the top (which costs
        // 3*NB live registers -- with 16 / 25 coefficients that alone pushed the backward to one wave per SIMD).
        float v[2][4][3];
        sh_load4<NATIVE, ALIGNED>(sh, hi, K, 0, v[0], mk);
#pragma unroll
        for (int k4 = 0; k4 < NV; ++k4) {
            if (k4 + 1 < NV) sh_load4<NATIVE, ALIGNED>(sh, hi, K, k4 + 1, v[(k4 + 1) & 1], mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) sh_accumulate<WITH_GRAD>(4 * k4 + i, dir, v[k4 & 1][i], col, Dx, Dy, Dz);
            asm volatile("" ::: "memory");
        }
    }
#pragma unroll
    for (int k = 4 * NV; k < NB; ++k) {
        const float v[3] = {sh_at<NATIVE>(sh, hi, K, k, 0, mk), sh_at<NATIVE>(sh, hi, K, k, 1, mk), sh_at<NATIVE>(sh, hi, K, k, 2, mk)};
        sh_accumulate<WITH_GRAD>(k, dir, v, col, Dx, Dy, Dz);
    }
}
// Direction gradient of the SH colour with dL/dcolour contracted FIRST:  dd = sum_k grad basis_k(dir) * (sh_k . g).
// Three accumulators (sh_contract<WITH_GRAD> carries sum_k grad basis_k sh_k[c] per channel -- nine -- plus the colour
// itself, because it learns which channels the forward clamped only at the end of the pass; here `g` already has those
// channels zeroed, from SpfState.sh_clamp).  Round 5: 255 -> 227 VGPRs at degree 3, 348 -> 256 at degree 4 (two waves
// per SIMD instead of one).
template <int NB, int NATIVE, bool ALIGNED>
__device__ __forceinline__ void sh_direction_gradient(const float* __restrict__ sh, const float* __restrict__ hi, int K,
                                                      const ShDir& dir, const float g[3], float dd[3], kfloat_p mk = nullptr) {
    constexpr int NV = NB / 4;
    auto acc = [&](int k, const float v[3]) {
        float gx, gy, gz;
        (void)sh_term<true>(k, dir, gx, gy, gz);
        const float sk = v[0] * g[0] + v[1] * g[1] + v[2] * g[2];
        if (sh_has_gx(k)) dd[0] += gx * sk;
        if (sh_has_gy(k)) dd[1] += gy * sk;
        if (sh_has_gz(k)) dd[2] += gz * sk;
    };
    // groups of four coefficients, double-buffered by hand behind a compiler barrier (see sh_contract)
    float v[2][4][3];
    if (NV > 0) sh_load4<NATIVE, ALIGNED>(sh, hi, K, 0, v[0], mk);
#pragma unroll
    for (int k4 = 0; k4 < NV; ++k4) {
        if (k4 + 1 < NV) sh_load4<NATIVE, ALIGNED>(sh, hi, K, k4 + 1, v[(k4 + 1) & 1], mk);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc(4 * k4 + i, v[k4 & 1][i]);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int k = 4 * NV; k < NB; ++k) {
        const float t[3] = {sh_at<NATIVE>(sh, hi, K, k, 0, mk), sh_at<NATIVE>(sh, hi, K, k, 1, mk), sh_at<NATIVE>(sh, hi, K, k, 2, mk)};
        acc(k, t);
    }
}
// dL/dsh_k = sum over the parked views of basis_k(direction_v) * dL/dcolour_v: evaluated group by group at the END of
// the view loop from six floats per view that every thread parks in LDS for itself (unit direction, colour gradient
// with clamped channels already zeroed) -- instead of 3*NB accumulator registers carried through the whole loop.
// park[(6 * v + j) * kBlock + tid]; `first`: store, otherwise add to what an earlier chunk of views stored.
// TO_LDS: `o` is the thread's own 3*K-float row of an LDS staging buffer (element-wise stores; `first` is true).
template <int NB, int NATIVE, bool ALIGNED, bool TO_LDS = false>
__device__ __forceinline__ void sh_grad_from_parked(float* __restrict__ o, float* __restrict__ o_hi, int K,
                                                    const float* __restrict__ park, int nviews, bool first,
                                                    kfloat_p mk = nullptr) {
    static_assert(!(TO_LDS && NATIVE == 2), "the split layout writes its planes directly");
    constexpr int NV = NB / 4;
    const int sk = NATIVE ? 1 : 3, sc = NATIVE ? K : 1;
    auto group = [&](int k0, int n, float (&acc)[4][3]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
        for (int v = 0; v < nviews; ++v) {
            const float* __restrict__ pv = park + (size_t)(6 * v) * kBlock;
            const ShDir dir = sh_dir(pv[0], pv[kBlock], pv[2 * kBlock]);
            const float g[3] = {pv[3 * kBlock], pv[4 * kBlock], pv[5 * kBlock]};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < n) {
                    float gx, gy, gz;
                    const float b = sh_term<false>(k0 + i, dir, gx, gy, gz);
                    acc[i][0] += b * g[0]; acc[i][1] += b * g[1]; acc[i][2] += b * g[2];
                }
        }
        if (NATIVE == 3) {       // raw rows: dL/draw = dL/dsh * sh_mask (what spf_adapter_backward does with dL/dsh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < n) { acc[i][0] *= mk[k0 + i]; acc[i][1] *= mk[k0 + i]; acc[i][2] *= mk[k0 + i]; }
        }
    };
#pragma unroll
    for (int k4 = 0; k4 < NV; ++k4) {
        float acc[4][3];
        group(4 * k4, 4, acc);
        if (TO_LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 4 * k4 + i;
                o[sk * k] = acc[i][0]; o[sk * k + sc] = acc[i][1]; o[sk * k + 2 * sc] = acc[i][2];
            }
            continue;
        }
        if (!first) {
            float old[4][3];
            sh_load4<(NATIVE == 3 ? 1 : NATIVE), ALIGNED>(o, o_hi, K, k4, old);      // (what an earlier chunk stored: already masked)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][0] += old[i][0]; acc[i][1] += old[i][1]; acc[i][2] += old[i][2]; }
        }
        sh_store4<NATIVE, ALIGNED>(o, o_hi, K, k4, acc);
    }
    if (NB % 4 != 0) {
        float acc[4][3];
        group(4 * NV, NB % 4, acc);
#pragma unroll
        for (int i = 0; i < NB % 4; ++i) {
            const int k = 4 * NV + i;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float* __restrict__ q = sh_slot<NATIVE>(o, o_hi, K, k, c);
                *q = first ? acc[i][c] : acc[i][c] + *q;
            }
        }
    }
    if (TO_LDS) {
        for (int k = NB; k < K; ++k) { o[sk * k] = 0.f; o[sk * k + sc] = 0.f; o[sk * k + 2 * sc] = 0.f; }
    } else if (first && K > NB) {     // coefficients that are carried but not evaluated: zero gradient
        if (NATIVE) {
#pragma unroll
            for (int c = 0; c < 3; ++c) zero_floats(o + c * K + NB, K - NB);
        } else {
            zero_floats(o + 3 * NB, 3 * (K - NB));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(G/256), S), block = 256.  DEG = -1: colours given (colors_precomp).
// ------------------------------------------------------------------------------------------
// minimum resident blocks per CU asked of the compiler (profiling knobs; 1 = no constraint)
#ifndef SPF_PBWD_BPC
#define SPF_PBWD_BPC 1
#endif
#ifndef SPF_PFWD_BPC
#define SPF_PFWD_BPC 1
#endif
#ifndef SPF_PBWD_DEG4_BPC
#define SPF_PBWD_DEG4_BPC 2   // (round 4: 348 VGPRs, the cap spilled 96 of them; round 5: 256 with the contracted direction gradient)
#endif
// Per-tile bookkeeping of a block: one packed word per (view of the group, tile) in LDS -- pairs in the top 12 bits
// (a block holds 256 Gaussians), footprint load (sum of cull-box areas capped at 256 each: <= 65,536) in the low 20.
constexpr uint32_t kHistCountShift = 20u, kHistAreaMask = (1u << 20) - 1u;
// views whose histograms fit the LDS budget at once (the loop flushes between groups)
// (direct bins: a group also holds the bins' base offsets [VG][T] and every thread's (rect, depth) per view)
__host__ __device__ inline int hist_view_group(int V, int T, bool direct = false) {
    const int fit = direct ? (36 * 1024) / (8 * T + 2048) : (32 * 1024) / (4 * T);
    return fit < 1 ? 1 : (fit < V ? fit : V);
}

// Direct bins (d.bin_cap > 0; lds_hist only): the kernel ALSO bins.  After the views of a group the block reserves,
// with one returning global atomic per touched (view, tile), a range of that tile's fixed bin (st.pairs[tile * bin_cap ..]),
// numbers its (Gaussian, tile) pairs from a sharded global cursor, and every thread writes its keys
// (depth bits << 32 | Gaussian) -- what spf_tile_scan_* + spf_bin_pairs_* did in two more launches and a second pass over
// rect / depth.  tile_count ends up as the bins' fill; nothing needs a scan.
template <int DEG, int NATIVE>
// (raw rows at degree 3 sit five registers above the 168 of three waves per SIMD: asked for, the compiler finds them)
#ifndef SPF_PFWD_RAW_BPC
#define SPF_PFWD_RAW_BPC 3
#endif
__global__ __launch_bounds__(kBlock, (NATIVE == 3 && DEG == 3) ? SPF_PFWD_RAW_BPC : SPF_PFWD_BPC) void spf_project_fwd_kernel(SpfDims d, SpfInputs in, SpfState st,
                                                                  int tiles_x, int tiles_y, int lds_hist) {
    // LDS: [VG][T] packed tile histograms of a group of views | [VG][4] per-wave pair totals | [4][64*12] record staging.
    // Nothing in the view loop waits for another wave: the histograms and the block's pair totals are flushed once per
    // group of views (normally: once), and a wave's 64 records leave through ITS staging buffer as full-wave contiguous
    // 16-byte stores (a lane's own three float4 stores at a 48-byte stride touch three times the cache lines).
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int T = tiles_x * tiles_y;
    const bool direct = d.bin_cap > 0;                                    // (host: only together with lds_hist)
    const int VG = lds_hist ? hist_view_group(d.V, T, direct) : d.V;
    uint32_t* const s_hist = s_dyn;                                       // [VG][T]   (lds_hist only)
    uint32_t* const s_wtot = s_dyn + (lds_hist ? VG * T : 0);             // [VG][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* const s_rec0 = reinterpret_cast<float4*>(s_wtot + ((VG * 4 + 3) & ~3));
    float4* const s_rec = s_rec0 + wave * (kWave * 3);
    // direct bins: [VG][T] base of the block's range in every bin | [VG][256] (rect, depth bits) of every thread | [VG] pair base
    uint32_t* const s_base = reinterpret_cast<uint32_t*>(s_rec0 + 4 * kWave * 3);
    uint2* const s_park = reinterpret_cast<uint2*>(s_base + VG * T);
    uint32_t* const s_vbase = reinterpret_cast<uint32_t*>(s_park + VG * kBlock);
    const int g = blockIdx.x * kBlock + threadIdx.x;
    const int s = blockIdx.y;
    const bool live = g < d.G;
    const size_t sg = (size_t)s * d.G + (live ? g : 0);
    const float p0[3] = {in.means3D[3 * sg], in.means3D[3 * sg + 1], in.means3D[3 * sg + 2]};
    // raw rows (sh_layout 3): the adapter's activations as the row is read -- the same expressions, in the same order,
    // as spf_adapter_fwd_kernel (adapter.hip)
    constexpr bool kRaw = NATIVE == 3;
    const float* __restrict__ raw_row = kRaw ? in.raw + sg * (size_t)d.raw_stride : nullptr;
    const kfloat_p mk = kRaw ? as_const(in.sh_mask) : nullptr;
    float sx, sy, sz;
    float4 q;
    if (kRaw) {
        sx = fminf(0.001f * softplus_torch(raw_row[0]), 0.3f) * d.scale_modifier;
        sy = fminf(0.001f * softplus_torch(raw_row[1]), 0.3f) * d.scale_modifier;
        sz = fminf(0.001f * softplus_torch(raw_row[2]), 0.3f) * d.scale_modifier;
        const float q0 = raw_row[3], q1 = raw_row[4], q2 = raw_row[5], q3 = raw_row[6];
        const float inv = 1.0f / (sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) + d.adapter_eps);
        q = make_float4(q0 * inv, q1 * inv, q2 * inv, q3 * inv);
    } else {
        sx = in.scales[3 * sg] * d.scale_modifier; sy = in.scales[3 * sg + 1] * d.scale_modifier;
        sz = in.scales[3 * sg + 2] * d.scale_modifier;
        q = *reinterpret_cast<const float4*>(in.rotations + 4 * sg);
    }
    const float opac = in.opacities[sg];
    float N0[9];
    {
        float R[9];
        quat_rot(q, R);
        scale_columns(R, sx, sy, sz, N0);
    }
    constexpr int NB = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
    if (lds_hist) {
        for (int t = threadIdx.x; t < VG * T; t += kBlock) s_hist[t] = 0;
        __syncthreads();
    }
    constexpr bool kLaneFocal = DEG <= 1;
    const LaneFocal lf = kLaneFocal ? lane_focal(in.tanfov, s, d.V, d.H, d.W, lane) : LaneFocal{0.f, 0.f};
    const int g_wave0 = blockIdx.x * kBlock + wave * kWave;              // first Gaussian of this wave
    const int n_wave = min(kWave, d.G - g_wave0);                        // its live Gaussians (<= 0: none)

    for (int v0 = 0; v0 < d.V; v0 += VG) {
      const int vend = min(d.V, v0 + VG);
      for (int v = v0; v < vend; ++v) {
        const int r = s * d.V + v;
        const kfloat_p Vm = as_const(in.viewmatrix + 16 * r), Pm = as_const(in.projmatrix + 16 * r);
        const kdouble_p M64 = in.viewmatrix64 ? as_const(in.viewmatrix64 + 16 * r) : nullptr;
        const float tanx = as_const(in.tanfov)[2 * r], tany = as_const(in.tanfov)[2 * r + 1];
        const float sc = in.view_scale ? as_const(in.view_scale)[r] : 1.0f;
        const float p[3] = {p0[0] * sc, p0[1] * sc, p0[2] * sc};

        Proj pr;
        float fx, fy;
        view_focal<kLaneFocal>(lf, d.V, v, tanx, tany, d.H, d.W, fx, fy);
        project_point(p, p0, Vm, M64, Pm, tanx, tany, fx, fy, d.H, d.W, N0, sc, pr);
        const float det = stable_det(pr);
        bool ok = live && pr.tz > kNearCull && det != 0.0f;
        float radius = 0.f;
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        float cA = 0.f, cB = 0.f, cC = 0.f;
        if (ok) {
            const float idet = 1.0f / det;
            cA = pr.c * idet; cB = -pr.b * idet; cC = pr.a * idet;
            // mid^2 - det = ((a - c)/2)^2 + b^2: the same quantity, again without the cancellation
            const float mid = 0.5f * (pr.a + pr.c), hd = 0.5f * (pr.a - pr.c);
            const float lam = mid + sqrtf(fmaxf(0.1f, hd * hd + pr.b * pr.b));
            radius = ceilf(3.0f * sqrtf(lam));
            ok = isfinite(pr.px) && isfinite(pr.py) && isfinite(radius);
            if (ok) {
                // The 3-sigma rect is part of the SEMANTICS: a Gaussian reaches exactly the pixels
                // of the 16 px tiles its radius box touches -- also pixels beyond the radius inside those tiles -- and is
                // culled (radii = 0) when it touches none (the cull-disc shrink below never drops a contributing pixel).
                constexpr int kSemTile = kTile;
                const float fgx = (float)((d.W + kSemTile - 1) / kSemTile), fgy = (float)((d.H + kSemTile - 1) / kSemTile);
                x0 = (int)fminf(fgx, fmaxf(0.f, truncf((pr.px - radius) * (1.0f / kSemTile))));
                y0 = (int)fminf(fgy, fmaxf(0.f, truncf((pr.py - radius) * (1.0f / kSemTile))));
                x1 = (int)fminf(fgx, fmaxf(0.f, truncf((pr.px + radius + (kSemTile - 1)) * (1.0f / kSemTile))));
                y1 = (int)fminf(fgy, fmaxf(0.f, truncf((pr.py + radius + (kSemTile - 1)) * (1.0f / kSemTile))));
                ok = (x1 - x0) * (y1 - y0) > 0;
            }
        }
        float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = make_float4(0.f, 0.f, 0.f, -1.f), rec2 = rec0;
        uint32_t rect_w = 0u, area = 0u;
        float zk = 0.f;
        int clamp_out = 0;                  // colour channels clamped at 0 (SpfState.sh_clamp: the backward's copy)
        if (ok) {
            // colour
            float col[3];
            int clampmask = 0;
            if (DEG < 0) {
                col[0] = in.colors[3 * sg]; col[1] = in.colors[3 * sg + 1]; col[2] = in.colors[3 * sg + 2];
            } else {
                // campos c_j = -sum_i t_i R[j][i]
                float dir[3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dir[j] = p[j] + (Vm[12] * Vm[4 * j] + Vm[13] * Vm[4 * j + 1] + Vm[14] * Vm[4 * j + 2]);
                const float inv = 1.0f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                const ShDir sd = sh_dir(dir[0] * inv, dir[1] * inv, dir[2] * inv);
                const float* __restrict__ sh = kRaw ? raw_row + 7 : in.shs + sg * (size_t)(NATIVE == 2 ? 16 : d.K) * 3;
                const float* __restrict__ sh_hi = NATIVE == 2 ? in.shs_high + sg * 27 : nullptr;
                col[0] = col[1] = col[2] = 0.f;
                if (NATIVE == 2 || (!kRaw && d.K % 4 == 0)) sh_contract<NB, NATIVE, true, false>(sh, sh_hi, d.K, sd, col, nullptr, nullptr, nullptr, mk);
                else sh_contract<NB, NATIVE, false, false>(sh, sh_hi, d.K, sd, col, nullptr, nullptr, nullptr, mk);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    col[ch] += 0.5f;
                    if (col[ch] < 0.f) { col[ch] = 0.f; clampmask |= 1 << ch; }
                }
            }
            // Conservative per-sub-tile cull radius: a pixel at squared distance d2 from the centre has
            // q = A dx^2 + 2B dx dy + C dy^2 >= mu_min * d2, and the pixel loop drops the Gaussian when
            // opacity * exp(-q/2) < 1/255, i.e. when q > 2 ln(255 * opacity).
            float cull_r2;
            const float thr = 2.0f * __logf(255.0f * opac);
            // (hardware square root and reciprocal, 1 ulp each: the 0.1 % slack below is four orders above that)
            const float mu = 0.5f * (cA + cC) - __builtin_amdgcn_sqrtf(0.25f * (cA - cC) * (cA - cC) + cB * cB);
            if (!(255.0f * opac > 1.0f)) cull_r2 = -1.0f;                       // never reaches 1/255
            else if (mu > 0.f) cull_r2 = (thr * 1.001f + 1e-3f) * __builtin_amdgcn_rcpf(mu) * 1.001f;  // slack for rounding
            else cull_r2 = 3.0e38f;
            if (!(cull_r2 == cull_r2)) cull_r2 = 3.0e38f;

            // Shrink the tile rect to the tiles that hold a pixel CENTRE inside the cull disc: a tile dropped here
            // could only have held pixels whose alpha is < 1/255, i.e. pairs that contribute nothing (the 3-sigma
            // rect above stays what `radii` reports).
            {
                const DiscBox db = disc_box_fast(pr.px, pr.py, cull_r2);
                if (!db.any) {
                    x1 = x0; y1 = y0;
                } else {
                    const float it = 1.0f / kTile;
                    x0 = max(x0, (int)fminf((float)tiles_x, fmaxf(0.f, floorf(db.xlo * it))));
                    y0 = max(y0, (int)fminf((float)tiles_y, fmaxf(0.f, floorf(db.ylo * it))));
                    x1 = min(x1, (int)fminf((float)tiles_x, fmaxf(0.f, floorf(db.xhi * it) + 1.f)));
                    y1 = min(y1, (int)fminf((float)tiles_y, fmaxf(0.f, floorf(db.yhi * it) + 1.f)));
                    if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; }
                }
            }
            rect_w = (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24);
            zk = pr.tz;
            rec0 = make_float4(pr.px, pr.py, cA, cB);
            rec1 = make_float4(cC, opac, pr.tz, cull_r2);
            rec2 = make_float4(col[0], col[1], col[2], __int_as_float(clampmask));
            clamp_out = clampmask;
            area = disc_area_capped_fast(pr.px, pr.py, cull_r2);
        }
        if (live) {
            // (per-lane 64-bit index as is: a scalar per-render base plus a 32-bit lane offset saves a few address
            //  instructions but measured 10 % SLOWER on the SH-heavy forwards -- K = 16: 139 -> 152 us, K = 25: 167 -> 187)
            const size_t rg_ = (size_t)r * d.G + g;
            st.radii[rg_] = ok ? (int)radius : 0;
            if (DEG >= 1) st.sh_clamp[rg_] = (uint8_t)clamp_out;            // (degree 0: the backward re-evaluates the one term)
            if (!direct) { st.rect[rg_] = rect_w; st.zkey[rg_] = zk; }      // (direct bins: parked below, binned by this block)
        }
        if (direct) s_park[(v - v0) * kBlock + threadIdx.x] = make_uint2(rect_w, __float_as_uint(zk));
        // ---- the wave's 64 records: own 48 bytes into LDS (conflict-free at this stride), contiguous 16-byte pieces out ----
        s_rec[3 * lane] = rec0; s_rec[3 * lane + 1] = rec1; s_rec[3 * lane + 2] = rec2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n_wave > 0) {
            float4* __restrict__ dst = reinterpret_cast<float4*>(st.rec + ((size_t)r * d.G + g_wave0) * kRec);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (lane + kWave * i < 3 * n_wave) dst[lane + kWave * i] = s_rec[lane + kWave * i];
        }
        __builtin_amdgcn_wave_barrier();                                   // (the buffer is rewritten by the next view)

        // ---- per-tile counts and footprint load ----
        const int npair = ok ? (x1 - x0) * (y1 - y0) : 0;
        if (lds_hist) {
            // A pixel-aligned scene puts runs of neighbouring lanes into the same tile.  Per-lane LDS atomics on one
            // address serialise (64 lanes: 64 passes); instead every run of equal tiles adds its total once: prefix sum
            // of the packed (1 pair, area) words over the lanes, run heads by a neighbour compare, and the last lane of
            // a run takes (prefix at its end) - (prefix before its head).  Gaussians with several tiles go one by one.
            uint32_t* __restrict__ hist = s_hist + (v - v0) * T;
            const bool single = npair == 1;
            const int tile = single ? y0 * tiles_x + x0 : -1;
            const uint32_t val = single ? ((1u << kHistCountShift) | area) : 0u;
            const uint32_t pre = wave_iscan_u32(val);
            const int prev_tile = dpp_wave_shr1(tile);
            const uint64_t heads = lane_ballot(tile != prev_tile) | 1ull;   // (no short-circuit around the DPP move:
                                                                            //  it must run with every lane active)
            const bool is_end = lane == kWave - 1 || ((heads >> (lane + 1)) & 1ull);
            const int head = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
            const uint32_t before = (uint32_t)__shfl((int)pre, max(head - 1, 0), kWave);
            if (is_end && single) atomicAdd(&hist[tile], pre - (head > 0 ? before : 0u));
            if (npair > 1) {
                const uint32_t one = (1u << kHistCountShift) | area;
                for (int ty = y0; ty < y1; ++ty)
                    for (int tx = x0; tx < x1; ++tx) atomicAdd(&hist[ty * tiles_x + tx], one);
            }
        } else if (npair > 0) {
            uint32_t* __restrict__ cnt = st.tile_count + (size_t)r * T;
            uint32_t* __restrict__ are = st.tile_flags + (size_t)r * T;
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    atomicAdd(&cnt[ty * tiles_x + tx], 1u);
                    atomicAdd(&are[ty * tiles_x + tx], area);
                }
        }
        // pairs produced by this wave for this render (feeds the Gaussian-major pair numbering)
        const uint32_t wtot = wave_iscan_u32((uint32_t)npair);
        if (lane == kWave - 1) s_wtot[(v - v0) * 4 + wave] = wtot;
      }
      if (direct) {
          // ---- direct bins: reserve, number, write keys (see the kernel's header comment) ----
          __syncthreads();
          const int nv = vend - v0;
          // (every global round trip of this tail is issued before any of them is waited for: the cursor's atomic by
          //  thread 0 and up to four bin reservations per thread go out together)
          uint32_t tot = 0, at = 0;
          uint64_t shard_base = 0, shard_cap = 0;
          if (threadIdx.x == 0) {
              // the block's pairs of this group of views, numbered from one cursor (Gaussian-major inside the block)
              for (int i = 0; i < nv; ++i) {
                  s_vbase[i] = tot;
                  tot += s_wtot[4 * i] + s_wtot[4 * i + 1] + s_wtot[4 * i + 2] + s_wtot[4 * i + 3];
              }
              const int nsh = pair_shards((int)(gridDim.x * gridDim.y));
              const int sh = (int)((blockIdx.x + blockIdx.y * gridDim.x) % (unsigned)nsh);
              shard_cap = (uint64_t)d.pair_capacity / (uint64_t)nsh;
              shard_base = shard_cap * (uint64_t)sh;
              if (tot) at = atomicAdd(&st.pair_cursor[sh], tot);
          }
          uint32_t longest = 0;
          for (int i0 = threadIdx.x; i0 < nv * T; i0 += 4 * kBlock) {
              uint32_t c[4], old[4] = {0u, 0u, 0u, 0u};
              size_t rt[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                  const int i = i0 + k * kBlock;
                  c[k] = i < nv * T ? s_hist[i] : 0u;
                  const int vi = i / T, t = i - vi * T;
                  rt[k] = (size_t)(s * d.V + v0 + vi) * T + t;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k)
                  if (c[k]) {
                      old[k] = atomicAdd(&st.tile_count[rt[k]], c[k] >> kHistCountShift);
                      atomicAdd(&st.tile_flags[rt[k]], c[k] & kHistAreaMask);
                  }
#pragma unroll
              for (int k = 0; k < 4; ++k)
                  if (c[k]) {
                      const int i = i0 + k * kBlock;
                      s_base[i] = old[k];
                      s_hist[i] = 0;                                                     // (now: slots handed out)
                      longest = max(longest, old[k] + (c[k] >> kHistCountShift));
                  }
          }
          if (threadIdx.x == 0) {
              if ((uint64_t)at + tot > shard_cap) {                                     // gradient records would not fit
                  atomicOr(&st.counters[2], 1u);
                  if (st.verdict_host) __hip_atomic_store(st.verdict_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
              s_vbase[VG] = (uint32_t)shard_base + at;
          }
          longest = wave_max_u32(longest);
          // (counters[1], the longest list, is NOT maintained with direct bins: one word visited by every wave of the launch
          //  -- 8,192 atomicMax, or even 8,192 write-through loads to look first -- serialises at ~11 ns each: 55 us / 180 us
          //  measured on a 52 us kernel.  The verdict the plan needs is local: a bin that overflows raises flag 2.)
          if (lane == 0 && longest > (uint32_t)d.bin_cap) {
              atomicOr(&st.counters[2], 2u);
              // (a host that wants the verdict early reads this host-mapped word behind an event: any non-zero value)
              if (st.verdict_host) __hip_atomic_store(st.verdict_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          __syncthreads();
          const uint32_t pbase = s_vbase[VG];
          for (int vi = 0; vi < nv; ++vi) {
              const int r = s * d.V + v0 + vi;
              const uint2 pk = s_park[vi * kBlock + threadIdx.x];
              const uint32_t rc = pk.x;
              const int x0 = rc & 0xff, y0 = (rc >> 8) & 0xff, x1 = (rc >> 16) & 0xff, y1 = rc >> 24;
              const uint32_t cnt = (x1 > x0 && y1 > y0) ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
              const uint32_t inc = wave_iscan_u32(cnt);
              uint32_t off = pbase + s_vbase[vi] + inc - cnt;
              for (int w = 0; w < wave; ++w) off += s_wtot[4 * vi + w];
              if (live) reinterpret_cast<uint2*>(st.pair_off)[(size_t)r * d.G + g] = make_uint2(rc, off);
              const uint64_t key = ((uint64_t)pk.y << 32) | (uint32_t)g;
              uint32_t* __restrict__ slots = s_hist + vi * T;
              const uint32_t* __restrict__ bs = s_base + vi * T;
              uint64_t* __restrict__ bins = st.pairs + (size_t)r * T * (size_t)d.bin_cap;
              const bool single = cnt == 1u;
              const int stile = single ? y0 * tiles_x + x0 : -1;
              const LaneRun run = lane_runs(stile, lane);
              uint32_t first = 0u;
              if (single && run.head == lane) first = atomicAdd(&slots[stile], (uint32_t)run.len);
              first = (uint32_t)__shfl((int)first, run.head, kWave);                    // (uniform flow: every lane)
              if (single) {
                  const uint32_t pos = bs[stile] + first + (uint32_t)(lane - run.head);
                  if (pos < (uint32_t)d.bin_cap) bins[(size_t)stile * d.bin_cap + pos] = key;
              }
              if (cnt > 1u) {
                  // several tiles: four slot requests in flight (one at a time, every tile pays the LDS atomic's round
                  // trip and the wave runs as many of them as its largest rect has tiles)
                  int tx = x0, ty = y0;
                  for (uint32_t i0 = 0; i0 < cnt; i0 += 4) {
                      int tt[4];
                      uint32_t pos[4];
#pragma unroll
                      for (int k = 0; k < 4; ++k) {
                          tt[k] = ty * tiles_x + tx;
                          pos[k] = (uint32_t)d.bin_cap;
                          if (i0 + k < cnt) pos[k] = atomicAdd(&slots[tt[k]], 1u);
                          if (++tx == x1) { tx = x0; ++ty; }
                      }
#pragma unroll
                      for (int k = 0; k < 4; ++k)
                          if (i0 + k < cnt) {
                              pos[k] += bs[tt[k]];
                              if (pos[k] < (uint32_t)d.bin_cap) bins[(size_t)tt[k] * d.bin_cap + pos[k]] = key;
                          }
                  }
              }
          }
          if (vend < d.V) {
              __syncthreads();
              for (int t = threadIdx.x; t < VG * T; t += kBlock) s_hist[t] = 0;
              __syncthreads();
          }
          continue;
      }
      // ---- flush the group: block pair totals and the tiles this block touched (one global atomic pair per tile) ----
      __syncthreads();
      for (int i = threadIdx.x; i < vend - v0; i += kBlock)
          st.blk_total[(size_t)(s * d.V + v0 + i) * gridDim.x + blockIdx.x] =
              s_wtot[4 * i] + s_wtot[4 * i + 1] + s_wtot[4 * i + 2] + s_wtot[4 * i + 3];
      if (lds_hist) {
          for (int i = threadIdx.x; i < (vend - v0) * T; i += kBlock) {
              const uint32_t c = s_hist[i];
              if (c) {
                  const int vi = i / T, t = i - vi * T;
                  const size_t rt = (size_t)(s * d.V + v0 + vi) * T + t;
                  atomicAdd(&st.tile_count[rt], c >> kHistCountShift);
                  atomicAdd(&st.tile_flags[rt], c & kHistAreaMask);
                  s_hist[i] = 0;
              }
          }
      }
      if (vend < d.V) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Backward.  Same decomposition; per-view viewmatrix partials are reduced per block into
// vpartial[r][block][12] (no float atomics -> deterministic), summed by spf_view_reduce_kernel.
// ------------------------------------------------------------------------------------------
constexpr int kViewChunk = 64;
constexpr int kShChunk = 8;
// dL/dsh leaves through an LDS staging buffer (see the backward kernel) when a Gaussian's coefficient block is not a
// multiple of 16 bytes per channel row (K = 25: every 16-byte piece is misaligned and straddles sectors; with
// K = 16 the direct stores are as fast: measured), all views fit one parked chunk and the block still fits twice on a
// CU (<= 80 KB of LDS)
// (raw rows, sh_layout 3: the staged row is the whole dL_draw row, 7 + 3K floats, and staging does not depend on K % 4)
__host__ __device__ inline int sh_stage_row(int K, bool raw) { return raw ? 7 + 3 * K : 3 * K; }
__host__ __device__ inline bool sh_stage_out(int V, int K, bool raw = false) {
    return (raw || K % 4 != 0) && V <= kShChunk &&
           (size_t)V * (6 * 256 + 48) * 4 + (size_t)128 * sh_stage_row(K, raw) * 4 <= 80 * 1024;
}      // views parked per thread before their SH gradient is formed (6 floats each, in LDS)

// (degree >= 2: 9..25 coefficients per channel.  Left alone the scheduler hoists every coefficient load to the top of
// the SH section -- 300+ VGPRs, one wave per SIMD; asking for two blocks per CU caps it at 256 VGPRs)
template <int DEG, int NATIVE>
__global__ __launch_bounds__(kBlock, ((DEG == 2 || DEG == 3) ? 2 : (DEG == 4 ? SPF_PBWD_DEG4_BPC : SPF_PBWD_BPC))) void spf_project_bwd_kernel(SpfDims d, SpfInputs in, SpfState st,
                                                                  SpfGrads gr, int nblk, uint64_t capacity) {
    (void)capacity;
    const int g = blockIdx.x * kBlock + threadIdx.x;
    const int s = blockIdx.y;
    const bool live = g < d.G;
    const size_t sg = (size_t)s * d.G + (live ? g : 0);
    if (st.counters[2] != 0u) {
        // the forward's plan did not hold (see render.hip::poison_tile): nothing was rendered and there are no pair
        // records -- every gradient this kernel owns becomes NaN instead of staying uninitialised
        const float nan = __builtin_nanf("");
        if (gr.vpartial)
            for (int i = threadIdx.x; i < d.V * 12; i += kBlock)
                gr.vpartial[((size_t)(s * d.V + i / 12) * nblk + blockIdx.x) * 12 + i % 12] = nan;
        else if (gr.dL_dviewmatrix)       // (a caller that reduces the partials itself: poison the result directly)
            for (int i = threadIdx.x; i < d.V * 16; i += kBlock) gr.dL_dviewmatrix[(size_t)s * d.V * 16 + i] = nan;
        if (!live) return;
        if (gr.dL_dmeans2D)
            for (int v = 0; v < d.V; ++v)
                for (int k = 0; k < 3; ++k) gr.dL_dmeans2D[((size_t)(s * d.V + v) * d.G + g) * 3 + k] = nan;
        for (int k = 0; k < 3; ++k) gr.dL_dmeans3D[3 * sg + k] = nan;
        gr.dL_dopacities[sg] = nan;
        if (gr.dL_dcolors) for (int k = 0; k < 3; ++k) gr.dL_dcolors[3 * sg + k] = nan;
        if (NATIVE == 3) {
            for (int k = 0; k < 7 + 3 * d.K; ++k) gr.dL_draw[sg * (size_t)(7 + 3 * d.K) + k] = nan;
            return;
        }
        if (gr.dL_dshs) {
            const int kk = NATIVE == 2 ? 16 : d.K;
            for (int k = 0; k < 3 * kk; ++k) gr.dL_dshs[sg * (size_t)kk * 3 + k] = nan;
            if (NATIVE == 2) for (int k = 0; k < 27; ++k) gr.dL_dshs_high[sg * 27 + k] = nan;
        }
        if (gr.dL_dscales) for (int k = 0; k < 3; ++k) gr.dL_dscales[3 * sg + k] = nan;
        if (gr.dL_drotations) for (int k = 0; k < 4; ++k) gr.dL_drotations[4 * sg + k] = nan;
        return;
    }
    float p0[3] = {0.f, 0.f, 0.f};
    float sx = 1.f, sy = 1.f, sz = 1.f, opac = 0.f;
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    constexpr bool kRaw = NATIVE == 3;
    const float* __restrict__ raw_row = kRaw ? in.raw + sg * (size_t)d.raw_stride : nullptr;
    const kfloat_p mk = kRaw ? as_const(in.sh_mask) : nullptr;
    const int CR = 7 + 3 * d.K;                                       // floats of a raw row / of a dL_draw row
    float raw_dact[kRaw ? 3 : 1] = {};                                // (raw rows) d scale / d raw[0:3] without the 0.001
    if (live) {
        p0[0] = in.means3D[3 * sg]; p0[1] = in.means3D[3 * sg + 1]; p0[2] = in.means3D[3 * sg + 2];
        if (kRaw) {      // (the adapter's activations, as in the forward kernel)
            float act[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // ... and what their backward needs, formed here once (three registers through the view loop instead of
                // three more softplus + exp chains and a second look at the row at the end): softplus' x the clamp's pass
                const float xx = raw_row[i];
                const float sp = softplus_torch(xx);
                act[i] = fminf(0.001f * sp, 0.3f) * d.scale_modifier;
                const float dsp = xx > 20.f ? 1.f : 1.f / (1.f + expf(-xx));
                raw_dact[i] = (0.001f * sp <= 0.3f) ? dsp : 0.f;
            }
            sx = act[0]; sy = act[1]; sz = act[2];
            const float q0 = raw_row[3], q1 = raw_row[4], q2 = raw_row[5], q3 = raw_row[6];
            const float inv = 1.0f / (sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) + d.adapter_eps);
            q = make_float4(q0 * inv, q1 * inv, q2 * inv, q3 * inv);
        } else {
            sx = in.scales[3 * sg] * d.scale_modifier; sy = in.scales[3 * sg + 1] * d.scale_modifier;
            sz = in.scales[3 * sg + 2] * d.scale_modifier;
            q = *reinterpret_cast<const float4*>(in.rotations + 4 * sg);
        }
        opac = in.opacities[sg];
    }
    (void)opac;
    float N0[9];
    {
        float R[9];
        quat_rot(q, R);
        scale_columns(R, sx, sy, sz, N0);
    }
    constexpr int NB = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);

    float dp0[3] = {0.f, 0.f, 0.f};          // dL/dmean3D
    float dN0[9];                            // dL/dN, N = R diag(s) (summed over the views)
#pragma unroll
    for (int k = 0; k < 9; ++k) dN0[k] = 0.f;
    float dopac = 0.f;
    float dcol[3] = {0.f, 0.f, 0.f};        // colours given directly

    extern __shared__ float s_part[];        // [min(V, kViewChunk)][4 waves][12]: viewmatrix partials of a chunk of views
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr bool kLaneFocal = DEG <= 1;
    const LaneFocal lf = kLaneFocal ? lane_focal(in.tanfov, s, d.V, d.H, d.W, lane) : LaneFocal{0.f, 0.f};
    // SH only: six floats per view that every thread parks for ITSELF (no barrier): see sh_grad_from_parked
    float* __restrict__ s_park = s_part + (d.V < kViewChunk ? d.V : kViewChunk) * 48 + threadIdx.x;
    constexpr bool kPark = DEG >= 2;         // few coefficients (K = 1, 4): plain register accumulators are cheaper
    const bool want_dsh = DEG >= 0 && (kRaw ? gr.dL_draw != nullptr : gr.dL_dshs != nullptr);
    float* __restrict__ dsh_out = !want_dsh ? nullptr : kRaw ? gr.dL_draw + sg * (size_t)CR + 7
                                                             : gr.dL_dshs + sg * (size_t)(NATIVE == 2 ? 16 : d.K) * 3;
    float* __restrict__ dsh_out_hi = (want_dsh && NATIVE == 2) ? gr.dL_dshs_high + sg * 27 : nullptr;
    // When every view fits the parked chunk, dL/dsh leaves through an LDS staging buffer (half a block at a time) and
    // is written with full-wave contiguous stores.  Per-thread stores of a 3*K-float block are 16-byte pieces at a
    // 12*K-byte stride: lines fill up piece by piece over the whole flush and the set of open lines outgrows the L2
    // -- at K = 25 the stores alone were 250 of the kernel's 439 us.
    const bool stage_out = NATIVE != 2 && kPark && want_dsh && sh_stage_out(d.V, d.K, kRaw);    // (split planes: direct stores)
    float dsh[kPark ? 1 : NB][3];
#pragma unroll
    for (int k = 0; k < (kPark ? 1 : NB); ++k) dsh[k][0] = dsh[k][1] = dsh[k][2] = 0.f;

    // Software pipeline over the views.  A view's gradient records hang off two dependent global reads (rect /
    // pair_off -> records); issued inside the view they cost two memory round trips per view with three waves per SIMD to
    // hide them (24 of 72 us on the bench step).  So: the (rect, pair_off) words of view v+2 and the FIRST record of
    // view v+1 (93 % of the visible Gaussians touch one tile) are in flight while view v is chained.
    const int gs = grec_floats(gr.dL_ddepth != nullptr);       // packed records: 9 or 10 floats
    const size_t rg0 = (size_t)s * d.V * d.G + (live ? g : 0);
    auto pairs_of = [](uint32_t rc) -> int {
        return (int)(((rc >> 16) & 0xff) - (rc & 0xff)) * (int)((rc >> 24) - ((rc >> 8) & 0xff));
    };
    const uint2* __restrict__ pinfo = reinterpret_cast<const uint2*>(st.pair_off);     // (rect, first pair): one 8-byte load
    uint32_t rc_cur = 0u, po_cur = 0u, rc_nxt = 0u, po_nxt = 0u;
    if (live) { const uint2 pi = pinfo[rg0]; rc_cur = pi.x; po_cur = pi.y; }
    if (live && d.V > 1) { const uint2 pi = pinfo[rg0 + d.G]; rc_nxt = pi.x; po_nxt = pi.y; }
    f4a q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
    float q8 = 0.f, q9 = 0.f;
    if (pairs_of(rc_cur) > 0) {
        const float* __restrict__ gp = gr.gpair + (size_t)po_cur * gs;
        q0 = *reinterpret_cast<const f4u*>(gp); q1 = *reinterpret_cast<const f4u*>(gp + 4); q8 = gp[8];
        if (gs == 10) q9 = gp[9];
    }

    for (int v = 0; v < d.V; ++v) {
        const int r = s * d.V + v;
        const kfloat_p Vm = as_const(in.viewmatrix + 16 * r), Pm = as_const(in.projmatrix + 16 * r);
        const kdouble_p M64 = in.viewmatrix64 ? as_const(in.viewmatrix64 + 16 * r) : nullptr;
        const float tanx = as_const(in.tanfov)[2 * r], tany = as_const(in.tanfov)[2 * r + 1];
        const size_t rg = (size_t)r * d.G + (live ? g : 0);
        float fx, fy;                                  // (cross-lane read: in uniform control flow, see lane_focal)
        view_focal<kLaneFocal>(lf, d.V, v, tanx, tany, d.H, d.W, fx, fy);
        float dV[12];  // dL/dVm[4i+j] for i<3 (index 3i+j) and dL/dVm[12+j] (index 9+j)
#pragma unroll
        for (int k = 0; k < 12; ++k) dV[k] = 0.f;

        // a Gaussian without (Gaussian, tile) pairs in this view (culled: rect == 0, or no pixel centre in its
        // cull disc) received no gradient record -> nothing to chain
        const uint32_t rc = rc_cur, po = po_cur;
        const int npair = pairs_of(rc);
        const bool vis = npair > 0;
        // this view's first record (loaded one view ago) ...
        float4 g0 = make_float4(q0.x, q0.y, q0.z, q0.w), g1 = make_float4(q1.x, q1.y, q1.z, q1.w);
        float4 g2 = make_float4(q8, q9, 0.f, 0.f);
        // ... and the loads of the views behind it
        rc_cur = rc_nxt; po_cur = po_nxt;
        if (live && v + 2 < d.V) { const uint2 pi = pinfo[rg + 2 * (size_t)d.G]; rc_nxt = pi.x; po_nxt = pi.y; }
        q0 = f4a{0.f, 0.f, 0.f, 0.f}; q1 = q0; q8 = 0.f; q9 = 0.f;
        if (v + 1 < d.V && pairs_of(rc_cur) > 0) {
            const float* __restrict__ gp = gr.gpair + (size_t)po_cur * gs;
            q0 = *reinterpret_cast<const f4u*>(gp); q1 = *reinterpret_cast<const f4u*>(gp + 4); q8 = gp[8];
            if (gs == 10) q9 = gp[9];
        }
        float sh_x = 0.f, sh_y = 0.f, sh_z = 1.f, sh_g0 = 0.f, sh_g1 = 0.f, sh_g2 = 0.f;   // parked below (SH only)
        if (vis) {
            // the rest of this Gaussian's (Gaussian, tile) pairs: their screen-space gradient records add up
            {
                const float* __restrict__ gp = gr.gpair + (size_t)po * gs;
                for (int i = 1; i < npair; ++i) {
                    const f4a a0 = *reinterpret_cast<const f4u*>(gp + gs * i);
                    const f4a a1 = *reinterpret_cast<const f4u*>(gp + gs * i + 4);
                    g0.x += a0.x; g0.y += a0.y; g0.z += a0.z; g0.w += a0.w;
                    g1.x += a1.x; g1.y += a1.y; g1.z += a1.z; g1.w += a1.w;
                    g2.x += gp[gs * i + 8];
                    if (gs == 10) g2.y += gp[gs * i + 9];
                }
            }
            const float sc = in.view_scale ? as_const(in.view_scale)[r] : 1.0f;
            const float p[3] = {p0[0] * sc, p0[1] * sc, p0[2] * sc};
            float dp[3] = {0.f, 0.f, 0.f};
            // (fields 2..4 are dL/d(a, b, c) of the 2-D covariance itself: the render backward forms them per pixel
            //  from v = conic * offset, see spf_common.h -- no conic -> covariance step here)
            const float gx = g0.x, gy = g0.y, da = g0.z, db = g0.w, dc = g1.x;
            const float gdepth = g2.y;
            float gcol[3] = {g1.z, g1.w, g2.x};
            dopac += g1.y;
            if (gr.dL_dmeans2D) {
                float* m2 = gr.dL_dmeans2D + rg * 3;
                m2[0] = gx * 0.5f * d.W; m2[1] = gy * 0.5f * d.H; m2[2] = 0.f;
            }
            // ---- colour (first: while the SH working set is live none of the projection state below is yet) ----
            if (DEG < 0) {
                dcol[0] += gcol[0]; dcol[1] += gcol[1]; dcol[2] += gcol[2];
            } else {
                float vdir[3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    vdir[j] = p[j] + (Vm[12] * Vm[4 * j] + Vm[13] * Vm[4 * j + 1] + Vm[14] * Vm[4 * j + 2]);
                const float inv = 1.0f / sqrtf(vdir[0] * vdir[0] + vdir[1] * vdir[1] + vdir[2] * vdir[2]);
                const float x = vdir[0] * inv, y = vdir[1] * inv, z = vdir[2] * inv;
                const ShDir sd = sh_dir(x, y, z);
                const float* __restrict__ sh = kRaw ? raw_row + 7 : in.shs + sg * (size_t)(NATIVE == 2 ? 16 : d.K) * 3;
                const float* __restrict__ sh_hi = NATIVE == 2 ? in.shs_high + sg * 27 : nullptr;
                float dd[3] = {0.f, 0.f, 0.f};
                if (DEG == 0) {
                    // one term: re-evaluate the colour exactly as the forward kernel does (a colour clamped at 0 passes
                    // no gradient)
                    float col[3] = {0.f, 0.f, 0.f};
                    if (!kRaw && d.K % 4 == 0) sh_contract<NB, NATIVE, true, false>(sh, sh_hi, d.K, sd, col, nullptr, nullptr, nullptr, mk);
                    else sh_contract<NB, NATIVE, false, false>(sh, sh_hi, d.K, sd, col, nullptr, nullptr, nullptr, mk);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
                        if (col[ch] + 0.5f < 0.f) gcol[ch] = 0.f;
                } else {
                    // the forward's clamp decision (one byte per (render, Gaussian)), then ONE pass over the coefficient
                    // block for the direction gradient, dL/dcolour contracted first (sh_direction_gradient)
                    const uint32_t cm = st.sh_clamp[rg];
                    if (cm & 1u) gcol[0] = 0.f;
                    if (cm & 2u) gcol[1] = 0.f;
                    if (cm & 4u) gcol[2] = 0.f;
                    if (NATIVE == 2 || (!kRaw && d.K % 4 == 0)) sh_direction_gradient<NB, NATIVE, true>(sh, sh_hi, d.K, sd, gcol, dd, mk);
                    else sh_direction_gradient<NB, NATIVE, false>(sh, sh_hi, d.K, sd, gcol, dd, mk);
                }
                sh_x = x; sh_y = y; sh_z = z; sh_g0 = gcol[0]; sh_g1 = gcol[1]; sh_g2 = gcol[2];
                if (!kPark) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        float t0, t1, t2;
                        const float bk = sh_term<false>(k, sd, t0, t1, t2);
                        dsh[k][0] += bk * gcol[0]; dsh[k][1] += bk * gcol[1]; dsh[k][2] += bk * gcol[2];
                    }
                }
                if (DEG > 0) {
                    const float dot = dd[0] * x + dd[1] * y + dd[2] * z;
                    float dv[3] = {(dd[0] - x * dot) * inv, (dd[1] - y * dot) * inv, (dd[2] - z * dot) * inv};
                    // v = p - c, c_j = -sum_i tvec_i Vm[4j+i]  ->  v_j = p_j + sum_i tvec_i Vm[4j+i]
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        dp[j] += dv[j];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            dV[9 + i] += dv[j] * Vm[4 * j + i];    // d/dtvec_i
                            dV[3 * j + i] += dv[j] * Vm[12 + i];   // d/dVm[4j+i]
                        }
                    }
                }
            }
            Proj pr;
            project_point(p, p0, Vm, M64, Pm, tanx, tany, fx, fy, d.H, d.W, N0, sc, pr);
            float dt[3] = {0.f, 0.f, gdepth};  // dL/dt (view space)

            // ---- pixel centre -> t (through the projection matrix) ----
            {
                const float ax = 0.5f * d.W * gx, ay = 0.5f * d.H * gy;
                const float dhx = ax * pr.pw, dhy = ay * pr.pw;
                const float dhw = -pr.pw * pr.pw * (ax * pr.homx + ay * pr.homy);
#pragma unroll
                for (int i = 0; i < 3; ++i) dt[i] += Pm[4 * i] * dhx + Pm[4 * i + 1] * dhy + Pm[4 * i + 3] * dhw;
            }
            // ---- cov2D = B B^T + 0.3 I, B = M (sc N0): a = b0.b0, b = b0.b1, c = b1.b1 ----
            const float* m0 = pr.m0; const float* m1 = pr.m1;
            float db0[3], db1[3], dm0[3], dm1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                db0[k] = sc * (2.f * da * pr.b0[k] + db * pr.b1[k]);       // (dL/db0, dL/db1) x the render's scale
                db1[k] = sc * (2.f * dc * pr.b1[k] + db * pr.b0[k]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dm0[j] = db0[0] * N0[3 * j] + db0[1] * N0[3 * j + 1] + db0[2] * N0[3 * j + 2];
                dm1[j] = db1[0] * N0[3 * j] + db1[1] * N0[3 * j + 1] + db1[2] * N0[3 * j + 2];
#pragma unroll
                for (int k = 0; k < 3; ++k) dN0[3 * j + k] += m0[j] * db0[k] + m1[j] * db1[k];
            }
            // M = J Wcv, Wcv[i][j] = Vm[4j+i]:  m0[j] = J00 Vm[4j] + J02 Vm[4j+2], m1[j] = J11 Vm[4j+1] + J12 Vm[4j+2]
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dJ00 += dm0[j] * Vm[4 * j];
                dJ02 += dm0[j] * Vm[4 * j + 2];
                dJ11 += dm1[j] * Vm[4 * j + 1];
                dJ12 += dm1[j] * Vm[4 * j + 2];
                dV[3 * j + 0] += dm0[j] * pr.J00;
                dV[3 * j + 1] += dm1[j] * pr.J11;
                dV[3 * j + 2] += dm0[j] * pr.J02 + dm1[j] * pr.J12;
            }
            {
                const float itz = pr.itz, itz2 = itz * itz, itz3 = itz2 * itz;
                if (pr.inx) dt[0] += -fx * itz2 * dJ02;
                if (pr.iny) dt[1] += -fy * itz2 * dJ12;
                dt[2] += -fx * itz2 * dJ00 - fy * itz2 * dJ11 + 2.f * fx * pr.tcx * itz3 * dJ02 +
                         2.f * fy * pr.tcy * itz3 * dJ12;
            }
            // ---- t = p R + tvec ----
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dp[i] += Vm[4 * i] * dt[0] + Vm[4 * i + 1] * dt[1] + Vm[4 * i + 2] * dt[2];
#pragma unroll
                for (int j = 0; j < 3; ++j) dV[3 * i + j] += p[i] * dt[j];
                dV[9 + i] += dt[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) dp0[i] += sc * dp[i];
        }
        if (kPark && want_dsh) {
            // park this view's direction and colour gradient (zeros when the Gaussian is not visible in it); every
            // kShChunk views -- normally once, after the loop -- turn the parked views into dL/dsh
            float* __restrict__ pk = s_park + (size_t)(6 * (v % kShChunk)) * kBlock;
            pk[0] = sh_x; pk[kBlock] = sh_y; pk[2 * kBlock] = sh_z;
            pk[3 * kBlock] = sh_g0; pk[4 * kBlock] = sh_g1; pk[5 * kBlock] = sh_g2;
            if (live && !stage_out && ((v + 1) % kShChunk == 0 || v + 1 == d.V)) {
                const int nv = v % kShChunk + 1;
                const bool first = v < kShChunk;
                if (NATIVE == 2 || (!kRaw && d.K % 4 == 0)) sh_grad_from_parked<NB, NATIVE, true>(dsh_out, dsh_out_hi, d.K, s_park, nv, first, mk);
                else sh_grad_from_parked<NB, NATIVE, false>(dsh_out, dsh_out_hi, d.K, s_park, nv, first, mk);
            }
        }
        // ---- wave totals of the 12 viewmatrix partials of this view (no barrier inside the view loop) ----
        if (gr.vpartial) {
            float tot[3];
            wave_sum12(dV, tot);                       // tot[j] = total of dV[4j + (lane & 3)], in every lane
            if (lane < 4) {
                float* __restrict__ sp = s_part + ((v & (kViewChunk - 1)) * 4 + wave) * 12 + lane;
                sp[0] = tot[0]; sp[4] = tot[1]; sp[8] = tot[2];
            }
            // the four wave totals are combined once per chunk of kViewChunk views (normally: once, after the loop)
            if (((v + 1) & (kViewChunk - 1)) == 0 || v + 1 == d.V) {
                const int v0 = v & ~(kViewChunk - 1);
                __syncthreads();
                for (int i = threadIdx.x; i < (v + 1 - v0) * 12; i += kBlock) {
                    const int vi = i / 12, k = i - 12 * vi;
                    const float* __restrict__ sp = s_part + vi * 48 + k;
                    gr.vpartial[((size_t)(s * d.V + v0 + vi) * nblk + blockIdx.x) * 12 + k] =
                        sp[0] + sp[12] + sp[24] + sp[36];
                }
                if (v + 1 < d.V) __syncthreads();
            }
        }
    }
    // raw rows: dL/draw[0..6] -- the scale and rotation gradients chained through the adapter's activations, with the
    // expressions (and their order) of spf_adapter_bwd_kernel: the row this kernel writes is the one the adapter's
    // backward would have produced from this kernel's dL/dscales, dL/drotations, dL/dsh
    float graw[kRaw ? 7 : 1];
    if constexpr (kRaw) {
        float rr[7] = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 3; i < 7; ++i) rr[i] = live ? raw_row[i] : rr[i];         // (the quaternion: four floats from the L2)
        const float nrm = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5] + rr[6] * rr[6]);
        const float dn = nrm + d.adapter_eps, inv = 1.0f / dn;
        const float4 q2 = make_float4(rr[3] * inv, rr[4] * inv, rr[5] * inv, rr[6] * inv);
        float R[9];
        quat_rot(q2, R);
        float ds[3], dR[9];
        const float svv[3] = {sx, sy, sz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sv = svv[k];
            ds[k] = (dN0[k] * R[k] + dN0[3 + k] * R[3 + k] + dN0[6 + k] * R[6 + k]) * d.scale_modifier;
#pragma unroll
            for (int i = 0; i < 3; ++i) dR[3 * i + k] = dN0[3 * i + k] * sv;
        }
        const float r = q2.x, x = q2.y, y = q2.z, z = q2.w;
        float dq[4];
        dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] -
                       2.f * x * dR[8]);
        dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] -
                       2.f * y * dR[8]);
        dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] +
                       y * dR[7]);
#pragma unroll
        for (int i = 0; i < 3; ++i) graw[i] = ds[i] * 0.001f * raw_dact[i];     // (= ds * 0.001 * softplus' * pass: pass is 0 or 1)
        const float dot = dq[0] * rr[3] + dq[1] * rr[4] + dq[2] * rr[5] + dq[3] * rr[6];
        const float kk = nrm > 0.f ? dot / (nrm * dn * dn) : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) graw[3 + i] = dq[i] / dn - rr[3 + i] * kk;
    }
    if constexpr (NATIVE != 2) if (stage_out) {
        const int row = sh_stage_row(d.K, kRaw);
        float* __restrict__ s_out = s_part + (d.V < kViewChunk ? d.V : kViewChunk) * 48 + d.V * 6 * kBlock;
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                  // (buffer free; the parked views are complete)
            if ((int)(threadIdx.x >> 7) == half && live) {
                float* __restrict__ mine = s_out + (threadIdx.x & 127) * row;
                if constexpr (kRaw) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) mine[i] = graw[i];
                    mine += 7;
                }
                if (!kRaw && d.K % 4 == 0) sh_grad_from_parked<NB, NATIVE, true, true>(mine, nullptr, d.K, s_park, d.V, true, mk);
                else sh_grad_from_parked<NB, NATIVE, false, true>(mine, nullptr, d.K, s_park, d.V, true, mk);
            }
            __syncthreads();
            const int g0 = blockIdx.x * kBlock + half * 128;
            const int nflt = min(128, d.G - g0) * row;        // (<= 0: nothing of this half exists)
            float* __restrict__ dst = (kRaw ? gr.dL_draw : gr.dL_dshs) + ((size_t)s * d.G + g0) * row;
            for (int i = 4 * threadIdx.x; i < nflt; i += 4 * kBlock) {
                if (i + 4 <= nflt) {
                    const float4 q = *reinterpret_cast<const float4*>(s_out + i);
                    *reinterpret_cast<f4u*>(dst + i) = f4a{q.x, q.y, q.z, q.w};
                } else {
                    for (int j = i; j < nflt; ++j) dst[j] = s_out[j];
                }
            }
        }
    }
    if (!live) return;
    gr.dL_dmeans3D[3 * sg] = dp0[0]; gr.dL_dmeans3D[3 * sg + 1] = dp0[1]; gr.dL_dmeans3D[3 * sg + 2] = dp0[2];
    gr.dL_dopacities[sg] = dopac;
    if (DEG < 0) {
        if (gr.dL_dcolors) {
            gr.dL_dcolors[3 * sg] = dcol[0]; gr.dL_dcolors[3 * sg + 1] = dcol[1]; gr.dL_dcolors[3 * sg + 2] = dcol[2];
        }
    } else if (!kPark && want_dsh) {
        const int sk = NATIVE ? 1 : 3, sc = NATIVE ? d.K : 1;
#pragma unroll
        for (int k = 0; k < (kPark ? 0 : NB); ++k) {
            const float m = kRaw ? mk[k] : 1.f;              // (raw rows: dL/draw = dL/dsh * sh_mask)
            dsh_out[sk * k] = dsh[k][0] * m; dsh_out[sk * k + sc] = dsh[k][1] * m; dsh_out[sk * k + 2 * sc] = dsh[k][2] * m;
        }
        if (d.K > NB) {
            if (NATIVE) {
                for (int c = 0; c < 3; ++c) zero_floats(dsh_out + c * d.K + NB, d.K - NB);
            } else {
                zero_floats(dsh_out + 3 * NB, 3 * (d.K - NB));
            }
        }
    }
    if constexpr (kRaw) {
        if (!stage_out && gr.dL_draw) {                       // (staged: the seven geometric slots left with the row)
            float* __restrict__ o = gr.dL_draw + sg * (size_t)CR;
#pragma unroll
            for (int i = 0; i < 7; ++i) o[i] = graw[i];
        }
        return;
    }
    if (gr.dL_dscales && gr.dL_drotations) {
        // N[i][k] = R[i][k] s_k.  R is rebuilt from the quaternion here rather than carried through the view loop (nine
        // registers; the compiler would otherwise keep the copy it made for N0 alive: hence the opaque quaternion)
        // (degree 3 runs at the 256-register cap with spills and degree 1 sat two registers above the 168 of three waves per SIMD:
        //  from degree 1 on the quaternion and the scales are READ AGAIN here --
        //  28 bytes per Gaussian from the L2 -- instead of being held through the view loop, seven registers)
        float4 q2 = q;
        float sv[3] = {sx, sy, sz};
        if (DEG >= 1) {
            q2 = *reinterpret_cast<const float4*>(in.rotations + 4 * sg);
            sv[0] = in.scales[3 * sg] * d.scale_modifier; sv[1] = in.scales[3 * sg + 1] * d.scale_modifier;
            sv[2] = in.scales[3 * sg + 2] * d.scale_modifier;
        }
        asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w));
        float R[9];
        quat_rot(q2, R);
        float ds[3], dR[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ds[k] = (dN0[k] * R[k] + dN0[3 + k] * R[3 + k] + dN0[6 + k] * R[6 + k]) * d.scale_modifier;
#pragma unroll
            for (int i = 0; i < 3; ++i) dR[3 * i + k] = dN0[3 * i + k] * sv[k];
        }
        gr.dL_dscales[3 * sg] = ds[0]; gr.dL_dscales[3 * sg + 1] = ds[1]; gr.dL_dscales[3 * sg + 2] = ds[2];
        const float r = q2.x, x = q2.y, y = q2.z, z = q2.w;
        float4 dq;
        dq.x = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        dq.y = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] -
                      2.f * x * dR[8]);
        dq.z = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] -
                      2.f * y * dR[8]);
        dq.w = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] +
                      y * dR[7]);
        *reinterpret_cast<float4*>(gr.dL_drotations + 4 * sg) = dq;
    }
}

// Sum vpartial[r][0..nblk) -> dL_dviewmatrix[r] (4x4, last column zero).  grid = R, block = 64.
__global__ void spf_view_reduce_kernel(const float* __restrict__ vpartial, float* __restrict__ dview, int nblk) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int b = lane; b < nblk; b += kWave) {
        const float* pp = vpartial + ((size_t)r * nblk + b) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += pp[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if (lane < 16) {
        const int i = lane >> 2, j = lane & 3;
        float v = 0.f;
        if (j < 3) {
            const int k = (i < 3) ? 3 * i + j : 9 + j;
#pragma unroll
            for (int kk = 0; kk < 12; ++kk)
                if (kk == k) v = acc[kk];
        }
        dview[16 * r + lane] = v;
    }
}

// ---- host-side launchers (called from api.hip) ---------------------------------------------
// degree the SH basis is evaluated to: band 4 (d_sh = 25) only on request, see SpfDims.sh_band4
static inline int sh_eval_degree(const SpfDims& d) {
    const int cap = d.sh_band4 ? 4 : 3;
    return d.sh_degree > cap ? cap : d.sh_degree;
}
template <int DEG, int NATIVE>
static void project_fwd_t(dim3 grid, size_t sm, hipStream_t stream, const SpfDims& d, const SpfInputs& in,
                          const SpfState& st, int tiles_x, int tiles_y, int lds) {
    spf_project_fwd_kernel<DEG, NATIVE><<<grid, dim3(kBlock), sm, stream>>>(d, in, st, tiles_x, tiles_y, lds);
}
template <int DEG, int NATIVE>
static void project_bwd_t(dim3 grid, hipStream_t stream, const SpfDims& d, const SpfInputs& in, const SpfState& st,
                          const SpfGrads& g, int nblk, uint64_t capacity) {
    size_t lds = (size_t)(d.V < kViewChunk ? d.V : kViewChunk) * 48 * sizeof(float);
    if (DEG >= 2 && (g.dL_dshs || g.dL_draw)) {
        lds += (size_t)(d.V < kShChunk ? d.V : kShChunk) * 6 * kBlock * sizeof(float);
        if (sh_stage_out(d.V, d.K, NATIVE == 3))      // staged dL/dsh (raw rows: the dL/draw rows), half a block at a time
            lds += (size_t)128 * sh_stage_row(d.K, NATIVE == 3) * sizeof(float);
        if (lds > 64 * 1024) {
            // more than the default 64 KB of dynamic LDS: opt in, once per device and instantiation
            static std::atomic<bool> attr_set[64];      // (per instantiation; idempotent attribute, see binning.hip)
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spf_project_bwd_kernel<DEG, NATIVE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
            }
        }
    }
    spf_project_bwd_kernel<DEG, NATIVE><<<grid, dim3(kBlock), lds, stream>>>(d, in, st, g, nblk, capacity);
}
// The band-split layout (sh_layout 2: planes [S,G,3,16] | [S,G,3,9]) below degree 4 IS the native layout with K = 16 on
// plane 0 -- 16-byte aligned coefficient rows, and the band-4 plane is neither read nor (backward) written.
static inline SpfDims dims_for_kernels(const SpfDims& d, int deg) {
    SpfDims k = d;
    if (d.sh_layout == 2 && deg < 4) { k.sh_layout = 1; k.K = 16; }
    return k;
}
#define SPF_DISPATCH_DEG(FN, ...)                                        \
    if (d.sh_layout == 3) {                                              \
        switch (deg) {                                                   \
            case 0: FN<0, 3>(__VA_ARGS__); break;                        \
            case 1: FN<1, 3>(__VA_ARGS__); break;                        \
            case 2: FN<2, 3>(__VA_ARGS__); break;                        \
            case 3: FN<3, 3>(__VA_ARGS__); break;                        \
            default: FN<4, 3>(__VA_ARGS__); break;                       \
        }                                                                \
    } else if (d.sh_layout == 2 && deg == 4) { FN<4, 2>(__VA_ARGS__); } else \
    switch (deg * 2 + (native ? 1 : 0)) {                                \
        case -2: case -1: FN<-1, false>(__VA_ARGS__); break;             \
        case 0: FN<0, false>(__VA_ARGS__); break;                        \
        case 1: FN<0, true>(__VA_ARGS__); break;                         \
        case 2: FN<1, false>(__VA_ARGS__); break;                        \
        case 3: FN<1, true>(__VA_ARGS__); break;                         \
        case 4: FN<2, false>(__VA_ARGS__); break;                        \
        case 5: FN<2, true>(__VA_ARGS__); break;                         \
        case 6: FN<3, false>(__VA_ARGS__); break;                        \
        case 7: FN<3, true>(__VA_ARGS__); break;                         \
        case 8: FN<4, false>(__VA_ARGS__); break;                        \
        default: FN<4, true>(__VA_ARGS__); break;                        \
    }

hipError_t launch_project_fwd(const SpfDims& d_in, const SpfInputs& in, const SpfState& st, int tiles_x, int tiles_y,
                              hipStream_t stream) {
    const int deg = in.colors ? -1 : sh_eval_degree(d_in);
    const SpfDims d = dims_for_kernels(d_in, deg);
    dim3 grid((d.G + kBlock - 1) / kBlock, d.S);
    const bool native = d.sh_layout != 0;
    const int T = tiles_x * tiles_y;
    const int lds = T <= max_lds_tiles() ? 1 : 0;
    const bool direct = d.bin_cap > 0;
    if (direct && (!lds || !st.pairs || !st.pair_cursor)) return hipErrorInvalidValue;   // (api.hip checks first)
    const int VG = lds ? hist_view_group(d.V, T, direct) : d.V;
    // packed tile histograms of a group of views | per-wave pair totals | record staging (4 waves x 64 x 48 bytes)
    // [| direct bins: range bases [VG][T] | parked (rect, depth) [VG][256] | pair bases [VG + 1]]
    size_t sm = sizeof(uint32_t) * ((lds ? (size_t)VG * T : 0) + (((size_t)VG * 4 + 3) & ~(size_t)3)) +
                (size_t)kBlock * kRec * sizeof(float);
    if (direct) sm += sizeof(uint32_t) * ((size_t)VG * T + 2 * (size_t)VG * kBlock + (((size_t)VG + 1 + 3) & ~(size_t)3));
    if (sm > 64 * 1024) return hipErrorInvalidValue;     // (V > ~3,000 views per scene without LDS histograms)
    SPF_DISPATCH_DEG(project_fwd_t, grid, sm, stream, d, in, st, tiles_x, tiles_y, lds)
    return hipGetLastError();
}

hipError_t launch_project_bwd(const SpfDims& d_in, const SpfInputs& in, const SpfState& st, const SpfGrads& g,
                              int nblk, uint64_t capacity, hipStream_t stream) {
    const int deg = in.colors ? -1 : sh_eval_degree(d_in);
    const SpfDims d = dims_for_kernels(d_in, deg);
    dim3 grid(nblk, d.S);
    const bool native = d.sh_layout != 0;
    SPF_DISPATCH_DEG(project_bwd_t, grid, stream, d, in, st, g, nblk, capacity)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (g.dL_dviewmatrix) {
        spf_view_reduce_kernel<<<d.S * d.V, kWave, 0, stream>>>(g.vpartial, g.dL_dviewmatrix, nblk);
        e = hipGetLastError();
    }
    return e;
}
#undef SPF_DISPATCH_DEG

}  // namespace spf
