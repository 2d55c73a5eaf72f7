// 3x3 singular value decomposition that follows LAPACK's sgesdd step by step, for host and device.
//
// Why: the head's SVD (models/poseMF_shapeGaussian_net.py:137, torch.svd on the CPU = MKL sgesdd) returns singular vectors
// whose SIGNS are not determined by the mathematics, and those signs are inputs of the child joints' MLPs (:126-130).  A
// device SVD is only a drop-in if it makes LAPACK's choices.  For a 3x3 matrix sgesdd (JOBZ = 'A', path 5) is a fixed,
// short sequence:   sgebd2  Householder bidiagonalisation  A = Q B P^T  (slarfg / slarf)
//                   sbdsdc  -> slasdq -> sbdsqr on the 3x3 upper bidiagonal B: implicit (zero-)shift QR sweeps with the
//                           convergence / direction / shift rules of LAPACK, 2x2 blocks by slasv2, shifts by slas2,
//                           rotations by slartg (the LAPACK 3.10 form: c >= 0, r carries the sign of f -- the form MKL
//                           uses; with the pre-3.10 form 20-30 % of the vectors come out with other signs), then
//                           negative singular values flipped (rows of V^T) and a descending sort
//                   sormbr  U = Q U_B,  V^T = V_B^T P^T
// Every routine below restates the published reference-LAPACK algorithm for n = 3 in fp32 with the same operation order
// (no fused multiply-add: contraction is switched off so that host and device builds round alike).  Measured against
// torch.svd (MKL) on 2 x 10^5 matrices I + sigma N(0,1), sigma in {0.05, 0.5, 2}, and on the golden F matrices: see
// DESIGN.md section 4 (sign agreement table) and tests/test_host_logic.py.
//
// Layout: f row-major 3x3; u, v row-major 3x3 with singular vectors in COLUMNS (torch.svd's U, V), s descending.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HPS_HD __host__ __device__ __forceinline__
#else
#define HPS_HD inline
#endif

namespace hps {
namespace gesdd3 {

#pragma clang fp contract(off)        // restored to the HIP default (fast) at the end of this header

constexpr float kEps = 5.9604644775390625e-8f;        // SLAMCH('Epsilon') = 2^-24
constexpr float kUnfl = 1.17549435e-38f;              // SLAMCH('Safe minimum')
constexpr float kRtMin = 1.08420217e-19f;             // sqrt(safmin)
constexpr float kRtMax = 1.30438179e+19f;             // sqrt(safmax / 2)

HPS_HD float sgn(float a, float b) { return copysignf(fabsf(a), b); }          // Fortran SIGN(a, b)

// SLAPY2: sqrt(x^2 + y^2) without unnecessary overflow
HPS_HD float slapy2(float x, float y) {
    const float xa = fabsf(x), ya = fabsf(y);
    const float w = fmaxf(xa, ya), z = fminf(xa, ya);
    if (z == 0.0f) return w;
    const float q = z / w;
    return w * sqrtf(1.0f + q * q);
}

// SNRM2 of one or two elements (scaled sum of squares)
HPS_HD float snrm2(const float* x, int n) {
    float scale = 0.0f, ssq = 1.0f;
    for (int i = 0; i < n; ++i) {
        if (x[i] != 0.0f) {
            const float a = fabsf(x[i]);
            if (scale < a) {
                const float q = scale / a;
                ssq = 1.0f + ssq * (q * q);
                scale = a;
            } else {
                const float q = a / scale;
                ssq = ssq + q * q;
            }
        }
    }
    return scale * sqrtf(ssq);
}

// SLARFG: elementary reflector H = I - tau [1; v] [1; v]^T with H [alpha; x] = [beta; 0].  alpha <- beta, x <- v.
HPS_HD float slarfg(int n, float& alpha, float* x) {
    if (n <= 1) return 0.0f;
    const float xnorm = snrm2(x, n - 1);
    if (xnorm == 0.0f) return 0.0f;
    const float beta = -sgn(slapy2(alpha, xnorm), alpha);
    const float tau = (beta - alpha) / beta;
    const float scal = 1.0f / (alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[i] = x[i] * scal;
    alpha = beta;
    return tau;
}

// SLARTG (LAPACK 3.10): plane rotation with c >= 0 and r = sign(d, f)
HPS_HD void slartg(float f, float g, float& c, float& s, float& r) {
    const float f1 = fabsf(f), g1 = fabsf(g);
    if (g == 0.0f) { c = 1.0f; s = 0.0f; r = f; }
    else if (f == 0.0f) { c = 0.0f; s = sgn(1.0f, g); r = g1; }
    else if (f1 > kRtMin && f1 < kRtMax && g1 > kRtMin && g1 < kRtMax) {
        const float d = sqrtf(f * f + g * g);
        c = f1 / d;
        r = sgn(d, f);
        s = g / r;
    } else {
        const float u = fminf(3.40282347e+38f, fmaxf(kUnfl, fmaxf(f1, g1)));
        const float fs = f / u, gs = g / u;
        const float d = sqrtf(fs * fs + gs * gs);
        c = fabsf(fs) / d;
        r = sgn(d, f);
        s = gs / r;
        r = r * u;
    }
}

// SLAS2: singular values of [[f, g], [0, h]]
HPS_HD void slas2(float f, float g, float h, float& ssmin, float& ssmax) {
    const float fa = fabsf(f), ga = fabsf(g), ha = fabsf(h);
    const float fhmn = fminf(fa, ha), fhmx = fmaxf(fa, ha);
    if (fhmn == 0.0f) {
        ssmin = 0.0f;
        if (fhmx == 0.0f) ssmax = ga;
        else {
            const float mx = fmaxf(fhmx, ga), mn = fminf(fhmx, ga);
            const float q = mn / mx;
            ssmax = mx * sqrtf(1.0f + q * q);
        }
    } else if (ga < fhmx) {
        const float as = 1.0f + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
        const float q = ga / fhmx, au = q * q;
        const float c = 2.0f / (sqrtf(as * as + au) + sqrtf(at * at + au));
        ssmin = fhmn * c;
        ssmax = fhmx / c;
    } else {
        const float au = fhmx / ga;
        if (au == 0.0f) {
            ssmin = (fhmn * fhmx) / ga;
            ssmax = ga;
        } else {
            const float as = 1.0f + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
            const float p = as * au, q = at * au;
            const float c = 1.0f / (sqrtf(1.0f + p * p) + sqrtf(1.0f + q * q));
            ssmin = (fhmn * c) * au;
            ssmin = ssmin + ssmin;
            ssmax = ga / (c + c);
        }
    }
}

// SLASV2: SVD of [[f, g], [0, h]]:  [csl snl; -snl csl] [f g; 0 h] [csr -snr; snr csr] = diag(ssmax, ssmin)
HPS_HD void slasv2(float f, float g, float h, float& ssmin, float& ssmax, float& snr, float& csr, float& snl, float& csl) {
    float ft = f, fa = fabsf(f), ht = h, ha = fabsf(h);
    int pmax = 1;
    const bool swap = ha > fa;
    if (swap) {
        pmax = 3;
        float t = ft; ft = ht; ht = t;
        t = fa; fa = ha; ha = t;
    }
    const float gt = g, ga = fabsf(g);
    float clt, crt, slt, srt;
    if (ga == 0.0f) {
        ssmin = ha; ssmax = fa; clt = 1.0f; crt = 1.0f; slt = 0.0f; srt = 0.0f;
    } else {
        bool gasmal = true;
        if (ga > fa) {
            pmax = 2;
            if (fa / ga < kEps) {
                gasmal = false;
                ssmax = ga;
                ssmin = ha > 1.0f ? fa / (ga / ha) : (fa / ga) * ha;
                clt = 1.0f; slt = ht / gt; srt = 1.0f; crt = ft / gt;
            }
        }
        if (gasmal) {
            const float d = fa - ha;
            float l = (d == fa) ? 1.0f : d / fa;
            const float m = gt / ft;
            float t = 2.0f - l;
            const float mm = m * m, tt = t * t;
            const float s = sqrtf(tt + mm);
            const float r = (l == 0.0f) ? fabsf(m) : sqrtf(l * l + mm);
            const float a = 0.5f * (s + r);
            ssmin = ha / a;
            ssmax = fa * a;
            if (mm == 0.0f) {
                if (l == 0.0f) t = sgn(2.0f, ft) * sgn(1.0f, gt);
                else t = gt / sgn(d, ft) + m / t;
            } else {
                t = (m / (s + t) + m / (r + l)) * (1.0f + a);
            }
            l = sqrtf(t * t + 4.0f);
            crt = 2.0f / l;
            srt = t / l;
            clt = (crt + srt * m) / a;
            slt = (ht / ft) * srt / a;
        }
    }
    if (swap) { csl = srt; snl = crt; csr = slt; snr = clt; }
    else { csl = clt; snl = slt; csr = crt; snr = srt; }
    float tsign;
    if (pmax == 1) tsign = sgn(1.0f, csr) * sgn(1.0f, csl) * sgn(1.0f, f);
    else if (pmax == 2) tsign = sgn(1.0f, snr) * sgn(1.0f, csl) * sgn(1.0f, g);
    else tsign = sgn(1.0f, snr) * sgn(1.0f, snl) * sgn(1.0f, h);
    ssmax = sgn(ssmax, tsign);
    ssmin = sgn(ssmin, tsign * sgn(1.0f, f) * sgn(1.0f, h));
}


#ifndef GEBD2V
#define GEBD2V 1
#endif
#ifndef ROTV
#define ROTV 0
#endif
#ifndef SROTV
#define SROTV 0
#endif
// y' = c*y - s*x ; x' = s*y + c*x   (slasr form: temp=y)
HPS_HD void rot_pair(int var, float c, float s, float& x, float& y) {
    const float t = y;
    float ny, nx;
    switch (var) {
        default:
        case 0: ny = c * t - s * x; nx = s * t + c * x; break;
        case 1: ny = __builtin_fmaf(c, t, -(s * x)); nx = __builtin_fmaf(s, t, c * x); break;
        case 2: ny = __builtin_fmaf(-s, x, c * t); nx = __builtin_fmaf(c, x, s * t); break;
        case 3: ny = __builtin_fmaf(c, t, -(s * x)); nx = __builtin_fmaf(c, x, s * t); break;
        case 4: ny = __builtin_fmaf(-s, x, c * t); nx = __builtin_fmaf(s, t, c * x); break;
    }
    y = ny; x = nx;
}

// SLASR(SIDE = 'L', PIVOT = 'V'): rotations j = 1..cnt-1 of rows (ll + j - 1, ll + j) of the 3x3 vt, forward or backward
HPS_HD void slasr_left(bool forward, int ll, int cnt, const float* c, const float* s, float* vt) {
    for (int q = 1; q < cnt; ++q) {
        const int j = forward ? q : cnt - q;
        const float ct = c[j - 1], st = s[j - 1];
        float* r0 = vt + (ll + j - 2) * 3;          // row ll + j - 1 (1-based)
        float* r1 = r0 + 3;
        for (int i = 0; i < 3; ++i) rot_pair(ROTV, ct, st, r0[i], r1[i]);
    }
}

// SLASR(SIDE = 'R', PIVOT = 'V'): the same on columns of the 3x3 u
HPS_HD void slasr_right(bool forward, int ll, int cnt, const float* c, const float* s, float* u) {
    for (int q = 1; q < cnt; ++q) {
        const int j = forward ? q : cnt - q;
        const float ct = c[j - 1], st = s[j - 1];
        const int c0 = ll + j - 2, c1 = c0 + 1;     // 0-based columns
        for (int i = 0; i < 3; ++i) rot_pair(ROTV, ct, st, u[i * 3 + c0], u[i * 3 + c1]);
    }
}

// SBDSQR for the 3x3 upper bidiagonal (d, e) with U = V^T = I on entry.  Arrays are used 1-based like the Fortran.
// Returns false if the iteration limit is hit (LAPACK: INFO > 0).
HPS_HD bool sbdsqr3(float* d0, float* e0, float* u, float* vt) {
    const int n = 3;
    float D[4] = {0.f, d0[0], d0[1], d0[2]};
    float E[4] = {0.f, e0[0], e0[1], 0.f};
    const float tolmul = 10.0f;                                 // max(10, min(100, eps^(-1/8))): eps^(-1/8) = 2^3 = 8
    const float tol = tolmul * kEps;
    float smax = 0.0f;
    for (int i = 1; i <= n; ++i) smax = fmaxf(smax, fabsf(D[i]));
    for (int i = 1; i < n; ++i) smax = fmaxf(smax, fabsf(E[i]));
    float sminl = 0.0f;
    float sminoa = fabsf(D[1]);
    if (sminoa != 0.0f) {
        float mu = sminoa;
        for (int i = 2; i <= n; ++i) {
            mu = fabsf(D[i]) * (mu / (mu + fabsf(E[i - 1])));
            sminoa = fminf(sminoa, mu);
            if (sminoa == 0.0f) break;
        }
    }
    sminoa = sminoa / sqrtf((float)n);
    const float thresh = fmaxf(tol * sminoa, 6.0f * n * n * kUnfl);
    const int maxit = 6 * n * n;
    int iter = 0, oldll = -1, oldm = -1, m = n, idir = 0;
    bool ok = true;
    for (;;) {
        if (m <= 1) break;
        if (iter > maxit) { ok = false; break; }
        // find the diagonal block to work on
        smax = fabsf(D[m]);
        int ll = 0;
        bool split = false;
        for (int lll = 1; lll <= m - 1; ++lll) {
            ll = m - lll;
            const float abss = fabsf(D[ll]), abse = fabsf(E[ll]);
            if (abse <= thresh) { split = true; break; }
            smax = fmaxf(smax, fmaxf(abss, abse));
        }
        if (split) {
            E[ll] = 0.0f;
            if (ll == m - 1) { m = m - 1; continue; }          // bottom singular value converged
        } else {
            ll = 0;
        }
        ll = ll + 1;
        if (ll == m - 1) {                                     // 2 x 2 block
            float sigmn, sigmx, sinr, cosr, sinl, cosl;
            slasv2(D[m - 1], E[m - 1], D[m], sigmn, sigmx, sinr, cosr, sinl, cosl);
            D[m - 1] = sigmx; E[m - 1] = 0.0f; D[m] = sigmn;
            float* r0 = vt + (m - 2) * 3;
            float* r1 = r0 + 3;
            for (int i = 0; i < 3; ++i) rot_pair(SROTV, cosr, sinr, r0[i], r1[i]);
            for (int i = 0; i < 3; ++i) rot_pair(SROTV, cosl, sinl, u[i * 3 + m - 2], u[i * 3 + m - 1]);
            m = m - 2;
            continue;
        }
        // n = 3: a block that is not 2 x 2 is the whole matrix (1 <= ll <= m - 2, m <= 3).  Saying so turns every index of the
        // sweeps below into a constant (the device compiler otherwise expands each D[i] / E[i] / row access into select chains).
        ll = 1;
        m = 3;
        if (ll > oldm || m < oldll) idir = (fabsf(D[ll]) >= fabsf(D[m])) ? 1 : 2;      // chase from the larger end
        // convergence tests
        bool again = false;
        if (idir == 1) {
            if (fabsf(E[m - 1]) <= fabsf(tol) * fabsf(D[m])) { E[m - 1] = 0.0f; continue; }
            float mu = fabsf(D[ll]);
            sminl = mu;
            for (int lll = ll; lll <= m - 1; ++lll) {
                if (fabsf(E[lll]) <= tol * mu) { E[lll] = 0.0f; again = true; break; }
                mu = fabsf(D[lll + 1]) * (mu / (mu + fabsf(E[lll])));
                sminl = fminf(sminl, mu);
            }
        } else {
            if (fabsf(E[ll]) <= fabsf(tol) * fabsf(D[ll])) { E[ll] = 0.0f; continue; }
            float mu = fabsf(D[m]);
            sminl = mu;
            for (int lll = m - 1; lll >= ll; --lll) {
                if (fabsf(E[lll]) <= tol * mu) { E[lll] = 0.0f; again = true; break; }
                mu = fabsf(D[lll]) * (mu / (mu + fabsf(E[lll])));
                sminl = fminf(sminl, mu);
            }
        }
        if (again) continue;
        oldll = ll;
        oldm = m;
        // shift
        float shift, r;
        if ((float)n * tol * (sminl / smax) <= fmaxf(kEps, 0.01f * tol)) {
            shift = 0.0f;
        } else {
            float sll;
            if (idir == 1) { sll = fabsf(D[ll]); slas2(D[m - 1], E[m - 1], D[m], shift, r); }
            else { sll = fabsf(D[m]); slas2(D[ll], E[ll], D[ll + 1], shift, r); }
            if (sll > 0.0f) {
                const float q = shift / sll;
                if (q * q < kEps) shift = 0.0f;
            }
        }
        iter = iter + m - ll;
        const int cnt = m - ll + 1;                             // 3 here (a 2x2 block never reaches this point)
        float w1[2], w2[2], w3[2], w4[2];
        if (shift == 0.0f) {
            if (idir == 1) {                                    // zero-shift sweep, top to bottom
                float cs = 1.0f, oldcs = 1.0f, sn = 0.0f, oldsn = 0.0f;
                for (int i = ll; i <= m - 1; ++i) {
                    slartg(D[i] * cs, E[i], cs, sn, r);
                    if (i > ll) E[i - 1] = oldsn * r;
                    slartg(oldcs * r, D[i + 1] * sn, oldcs, oldsn, D[i]);
                    w1[i - ll] = cs; w2[i - ll] = sn; w3[i - ll] = oldcs; w4[i - ll] = oldsn;
                }
                const float h = D[m] * cs;
                D[m] = h * oldcs;
                E[m - 1] = h * oldsn;
                slasr_left(true, ll, cnt, w1, w2, vt);
                slasr_right(true, ll, cnt, w3, w4, u);
                if (fabsf(E[m - 1]) <= thresh) E[m - 1] = 0.0f;
            } else {                                            // bottom to top
                float cs = 1.0f, oldcs = 1.0f, sn = 0.0f, oldsn = 0.0f;
                for (int i = m; i >= ll + 1; --i) {
                    slartg(D[i] * cs, E[i - 1], cs, sn, r);
                    if (i < m) E[i] = oldsn * r;
                    slartg(oldcs * r, D[i - 1] * sn, oldcs, oldsn, D[i]);
                    w1[i - ll - 1] = cs; w2[i - ll - 1] = -sn; w3[i - ll - 1] = oldcs; w4[i - ll - 1] = -oldsn;
                }
                const float h = D[ll] * cs;
                D[ll] = h * oldcs;
                E[ll] = h * oldsn;
                slasr_left(false, ll, cnt, w3, w4, vt);
                slasr_right(false, ll, cnt, w1, w2, u);
                if (fabsf(E[ll]) <= thresh) E[ll] = 0.0f;
            }
        } else {
            if (idir == 1) {                                    // shifted sweep, top to bottom
                float f = (fabsf(D[ll]) - shift) * (sgn(1.0f, D[ll]) + shift / D[ll]);
                float g = E[ll];
                for (int i = ll; i <= m - 1; ++i) {
                    float cosr, sinr, cosl, sinl;
                    slartg(f, g, cosr, sinr, r);
                    if (i > ll) E[i - 1] = r;
                    f = cosr * D[i] + sinr * E[i];
                    E[i] = cosr * E[i] - sinr * D[i];
                    g = sinr * D[i + 1];
                    D[i + 1] = cosr * D[i + 1];
                    slartg(f, g, cosl, sinl, r);
                    D[i] = r;
                    f = cosl * E[i] + sinl * D[i + 1];
                    D[i + 1] = cosl * D[i + 1] - sinl * E[i];
                    if (i < m - 1) {
                        g = sinl * E[i + 1];
                        E[i + 1] = cosl * E[i + 1];
                    }
                    w1[i - ll] = cosr; w2[i - ll] = sinr; w3[i - ll] = cosl; w4[i - ll] = sinl;
                }
                E[m - 1] = f;
                slasr_left(true, ll, cnt, w1, w2, vt);
                slasr_right(true, ll, cnt, w3, w4, u);
                if (fabsf(E[m - 1]) <= thresh) E[m - 1] = 0.0f;
            } else {                                            // bottom to top
                float f = (fabsf(D[m]) - shift) * (sgn(1.0f, D[m]) + shift / D[m]);
                float g = E[m - 1];
                for (int i = m; i >= ll + 1; --i) {
                    float cosr, sinr, cosl, sinl;
                    slartg(f, g, cosr, sinr, r);
                    if (i < m) E[i] = r;
                    f = cosr * D[i] + sinr * E[i - 1];
                    E[i - 1] = cosr * E[i - 1] - sinr * D[i];
                    g = sinr * D[i - 1];
                    D[i - 1] = cosr * D[i - 1];
                    slartg(f, g, cosl, sinl, r);
                    D[i] = r;
                    f = cosl * E[i - 1] + sinl * D[i - 1];
                    D[i - 1] = cosl * D[i - 1] - sinl * E[i - 1];
                    if (i > ll + 1) {
                        g = sinl * E[i - 2];
                        E[i - 2] = cosl * E[i - 2];
                    }
                    w1[i - ll - 1] = cosr; w2[i - ll - 1] = -sinr; w3[i - ll - 1] = cosl; w4[i - ll - 1] = -sinl;
                }
                E[ll] = f;
                if (fabsf(E[ll]) <= thresh) E[ll] = 0.0f;
                slasr_left(false, ll, cnt, w3, w4, vt);
                slasr_right(false, ll, cnt, w1, w2, u);
            }
        }
    }
    // make the singular values positive, then sort them into decreasing order (one transposition per vector)
    for (int i = 1; i <= n; ++i) {
        if (D[i] < 0.0f) {
            D[i] = -D[i];
            for (int k = 0; k < 3; ++k) vt[(i - 1) * 3 + k] = -vt[(i - 1) * 3 + k];
        }
    }
    for (int i = 1; i <= n - 1; ++i) {
        int isub = 1;
        float smin = D[1];
        for (int j = 2; j <= n + 1 - i; ++j) {
            if (D[j] <= smin) { isub = j; smin = D[j]; }
        }
        const int last = n + 1 - i;
        if (isub != last) {
            D[isub] = D[last];
            D[last] = smin;
            for (int k = 0; k < 3; ++k) {
                float t = vt[(isub - 1) * 3 + k]; vt[(isub - 1) * 3 + k] = vt[(last - 1) * 3 + k]; vt[(last - 1) * 3 + k] = t;
                t = u[k * 3 + isub - 1]; u[k * 3 + isub - 1] = u[k * 3 + last - 1]; u[k * 3 + last - 1] = t;
            }
        }
    }
    d0[0] = D[1]; d0[1] = D[2]; d0[2] = D[3];
    return ok;
}

// sgesdd('A') of the row-major 3x3 matrix f:  f = u diag(s) v^T.  Returns false on non-convergence or non-finite input
// (LAPACK: INFO != 0; torch raises) -- the outputs are then NaN.
static float* g_dbg = nullptr;
HPS_HD bool svd3(const float* f, float* u, float* s, float* v) {
    float a[9];
    float anrm = 0.0f;
    bool finite = true;
    for (int i = 0; i < 9; ++i) {
        a[i] = f[i];
        anrm = fmaxf(anrm, fabsf(f[i]));
        if (!(fabsf(f[i]) <= 3.40282347e+38f)) finite = false;
    }
    if (!finite) {
        const float nan = __builtin_nanf("");
        for (int i = 0; i < 9; ++i) { u[i] = nan; v[i] = nan; }
        s[0] = s[1] = s[2] = nan;
        return false;
    }
    // sgesdd scales matrices whose largest entry is outside [smlnum, bignum] = [9.1e-13, 1.1e12]
    float rescale = 1.0f;
    if (anrm > 0.0f && anrm < 9.09494702e-13f) rescale = 9.09494702e-13f / anrm;
    else if (anrm > 1.09951163e+12f) rescale = 1.09951163e+12f / anrm;
    if (rescale != 1.0f)
        for (int i = 0; i < 9; ++i) a[i] = a[i] * rescale;

    // ---- sgebd2: A = Q B P^T, reflector vectors kept in a (below the diagonal / right of the superdiagonal) ----
    float d[3], e[2], tauq[3], taup0;
    {   // i = 1: H(1) annihilates a(2:3, 1)
        float x[2] = {a[3], a[6]};
        tauq[0] = slarfg(3, a[0], x);
        d[0] = a[0];
        a[3] = x[0]; a[6] = x[1];
        // Rounding of the reflector applications below = MKL's (see the note above svd3): w = c_1 + (v_2 c_2 + v_3 c_3) with
        // every product and sum rounded, the rank-one update as ONE fused multiply-add per element.
        if (tauq[0] != 0.0f) {                                  // slarf 'L' on a(1:3, 2:3)
            const float vq[3] = {1.0f, x[0], x[1]};
            for (int j = 1; j < 3; ++j) {
                const float w = GEBD2V ? a[j] + (a[3 + j] * vq[1] + a[6 + j] * vq[2]) : (a[j] + a[3 + j] * vq[1]) + a[6 + j] * vq[2];
                const float t = -tauq[0] * w;
                for (int r = 0; r < 3; ++r) a[r * 3 + j] = GEBD2V ? __builtin_fmaf(vq[r], t, a[r * 3 + j]) : a[r * 3 + j] + vq[r] * t;
            }
        }
        // G(1) annihilates a(1, 3)
        float y[1] = {a[2]};
        taup0 = slarfg(2, a[1], y);
        e[0] = a[1];
        a[2] = y[0];
        if (taup0 != 0.0f) {                                    // slarf 'R' on a(2:3, 2:3): here the two-term w is fused as well
            const float vp[2] = {1.0f, y[0]};
            float w[2];
            for (int r = 1; r < 3; ++r) w[r - 1] = GEBD2V ? __builtin_fmaf(vp[1], a[r * 3 + 2], a[r * 3 + 1]) : a[r * 3 + 1] + vp[1] * a[r * 3 + 2];
            for (int j = 0; j < 2; ++j) {
                const float t = -taup0 * vp[j];
                for (int r = 1; r < 3; ++r) a[r * 3 + 1 + j] = GEBD2V ? __builtin_fmaf(w[r - 1], t, a[r * 3 + 1 + j]) : a[r * 3 + 1 + j] + w[r - 1] * t;
            }
        }
    }
    {   // i = 2: H(2) annihilates a(3, 2); G(2) is trivial
        float x[1] = {a[7]};
        tauq[1] = slarfg(2, a[4], x);
        d[1] = a[4];
        a[7] = x[0];
        if (tauq[1] != 0.0f) {                                  // slarf 'L' on a(2:3, 3)
            const float vq[2] = {1.0f, x[0]};
            const float w = a[5] + a[8] * vq[1];
            const float t = -tauq[1] * w;
            for (int r = 0; r < 2; ++r) a[(r + 1) * 3 + 2] = GEBD2V ? __builtin_fmaf(vq[r], t, a[(r + 1) * 3 + 2]) : a[(r + 1) * 3 + 2] + vq[r] * t;
        }
        e[1] = a[5];
    }
    d[2] = a[8];                                                // i = 3: nothing to annihilate
    tauq[2] = 0.0f;

    // ---- bidiagonal SVD ----
    float vt[9];
    for (int i = 0; i < 9; ++i) { u[i] = (i % 4 == 0) ? 1.0f : 0.0f; vt[i] = u[i]; }
    const bool ok = sbdsqr3(d, e, u, vt);

    if (g_dbg) { for (int i = 0; i < 9; ++i) g_dbg[i] = u[i]; g_dbg[9] = a[3]; g_dbg[10] = a[6]; g_dbg[11] = a[7]; g_dbg[12] = tauq[0]; g_dbg[13] = tauq[1]; }
    // ---- sormbr variants ----
#ifndef LW
#define LW 0
#endif
#ifndef LU
#define LU 0
#endif
#ifndef RW
#define RW 0
#endif
#ifndef RU
#define RU 0
#endif
#ifndef RT
#define RT 0
#endif
    if (tauq[1] != 0.0f) {
        const float vq[2] = {1.0f, a[7]};
        for (int j = 0; j < 3; ++j) {
            float w;
            if (LW == 2) w = __builtin_fmaf(u[6 + j], vq[1], u[3 + j]);
            else w = u[3 + j] + u[6 + j] * vq[1];
            const float t = -tauq[1] * w;
            if (LU == 2) { u[3 + j] = __builtin_fmaf(-tauq[1], w, u[3 + j]); u[6 + j] = __builtin_fmaf(vq[1], t, u[6 + j]); }
            else for (int r = 0; r < 2; ++r) u[(r + 1) * 3 + j] = LU ? __builtin_fmaf(vq[r], t, u[(r + 1) * 3 + j]) : u[(r + 1) * 3 + j] + vq[r] * t;
        }
    }
    if (tauq[0] != 0.0f) {
        const float vq[3] = {1.0f, a[3], a[6]};
        for (int j = 0; j < 3; ++j) {
            float w;
            if (LW == 0) w = (u[j] + u[3 + j] * vq[1]) + u[6 + j] * vq[2];
            else if (LW == 1) w = u[j] + (u[3 + j] * vq[1] + u[6 + j] * vq[2]);
            else w = __builtin_fmaf(u[6 + j], vq[2], __builtin_fmaf(u[3 + j], vq[1], u[j]));
            const float t = -tauq[0] * w;
            if (LU == 2) { u[j] = __builtin_fmaf(-tauq[0], w, u[j]); u[3 + j] = __builtin_fmaf(vq[1], t, u[3 + j]); u[6 + j] = __builtin_fmaf(vq[2], t, u[6 + j]); }
            else for (int r = 0; r < 3; ++r) u[r * 3 + j] = LU ? __builtin_fmaf(vq[r], t, u[r * 3 + j]) : u[r * 3 + j] + vq[r] * t;
        }
    }
    if (taup0 != 0.0f) {
        const float vp[2] = {1.0f, a[2]};
        float w[3];
        for (int r = 0; r < 3; ++r) w[r] = RW ? __builtin_fmaf(vp[1], vt[r * 3 + 2], vt[r * 3 + 1]) : vt[r * 3 + 1] + vp[1] * vt[r * 3 + 2];
        for (int j = 0; j < 2; ++j) {
            for (int r = 0; r < 3; ++r) {
                if (RT == 0) { const float t = -taup0 * vp[j]; vt[r * 3 + 1 + j] = RU ? __builtin_fmaf(w[r], t, vt[r * 3 + 1 + j]) : vt[r * 3 + 1 + j] + w[r] * t; }
                else { const float t = -taup0 * w[r]; vt[r * 3 + 1 + j] = RU ? __builtin_fmaf(t, vp[j], vt[r * 3 + 1 + j]) : vt[r * 3 + 1 + j] + t * vp[j]; }
            }
        }
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v[r * 3 + c] = vt[c * 3 + r];
    const float inv = 1.0f / rescale;
    s[0] = rescale != 1.0f ? d[0] * inv : d[0];
    s[1] = rescale != 1.0f ? d[1] * inv : d[1];
    s[2] = rescale != 1.0f ? d[2] * inv : d[2];
    if (!ok) {
        const float nan = __builtin_nanf("");
        for (int i = 0; i < 9; ++i) { u[i] = nan; v[i] = nan; }
        s[0] = s[1] = s[2] = nan;
    }
    return ok;
}

#pragma clang fp contract(fast)

}  // namespace gesdd3
}  // namespace hps
