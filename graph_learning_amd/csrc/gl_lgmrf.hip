// Laplacian-constrained GMRF (Egilmez GGL): block-coordinate descent on MI355X.
//
// Reference: graph_learn/lgmrf.py
//   nonnegative_qp_solver :19-71   -> ADMM kernels below (gl_nonneg_qp_solve)
//   LGMRF._initialize     :152-167 -> k_lg_prepare
//   LGMRF._ggl_fit        :169-226 -> lgmrf_fit (per-node loop)
//   LGMRF.fit             :228-235 -> gl_lgmrf_fit
//   relative_error, graph_learn/evaluation.py:8-25 -> gl_relative_error
//
// State on the device (n x n fp64, row-major): K (scaled input + H), Theta
// (laplacian_), C = Theta^-1 (inv_laplacian_).  Per node u the reference forms
// M = C[m,m] - c c'/c_uu (m = all nodes but u), solves a non-negative QP on it by
// ADMM with the explicit inverse inv(I + rho M[nz,nz]) and applies two rank-1
// corrections.  Nothing is gathered here: all vectors are length n with entry u
// (and masked-out entries) pinned to 0, and M is applied as C x - c (c.x)/c_uu.
//
// Two ways to apply inv(I + rho Q) inside the ADMM loop:
//   general path (any adj_mask): W = I + rho M on the active set is built densely and
//     inverted in place by Gauss-Jordan (SPD, no pivoting) -- O(n^3) per node, like the
//     reference, for small problems and masked priors;
//   fast path (default full mask): M = (Theta_mm)^-1, hence
//         inv(I + rho M) = I - rho (Theta_mm + rho I)^-1,
//     and (Theta_mm + rho I)^-1 = G_mm - g g'/G_uu with G = (Theta + rho I)^-1 kept up to
//     date by a rank-2 Woodbury correction per node (row/column u of Theta changes) and
//     re-computed from scratch once per cycle -- the explicit inverse of a node is then an
//     O(n^2) elementwise kernel instead of an O(n^3) factorisation.
// Both produce the reference's iterates up to rounding; the stopping rules are the
// reference's.
#include <map>
#include <vector>

#include "gl_common.h"
#include "gl_gridbar.h"
#include "gl_lanes.h"

namespace {

constexpr int VT = 1024;  // threads of the single-workgroup vector kernels

__device__ __forceinline__ double wsum(double v) { return wave_sum(v); }  // gl_lanes.h: DPP / permlane swaps
__device__ __forceinline__ double wmax(double v) { return wave_max(v); }  // DPP butterfly (a ds_bpermute one cost 0.25 us)

// The ADMM's stopping rule (lgmrf.py:66-69 with relative_error, evaluation.py:21-25: plain norm when the
// previous iterate is allclose to 0) from the squared norms.  Four square roots and two divisions in fp64
// are ~0.35 us for a lone wave; they are only taken when the squares do not already say "not yet".
__device__ __forceinline__ bool lg_rule_met(double dx2, double x2, double xm, double dz2, double z2, double zm, double tol) {
    const bool xs = xm <= 1e-8, zs = zm <= 1e-8;
    const double t2 = tol * tol * (1 + 1e-12);
    if (dx2 > t2 * (xs ? 1.0 : x2) || dz2 > t2 * (zs ? 1.0 : z2)) return false;  // sqrt(a) / sqrt(b) > tol for sure
    const double rx = xs ? sqrt(dx2) : sqrt(dx2) / sqrt(x2);
    const double rz = zs ? sqrt(dz2) : sqrt(dz2) / sqrt(z2);
    return rx < tol && rz < tol;
}

// lane j's value in every lane, through the scalar file
__device__ __forceinline__ double lg_bcast(double x, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}

// sums of four per-lane values over the wave, all four in every lane: the halving butterfly of gl_lanes.h
// (7 exchanges instead of 4 x 6) and four scalar broadcasts
__device__ __forceinline__ void lg_wsum4(double (&v)[4], int lane) {
    const double c = add_xor32(halfwave_rows4(v, lane));
    v[0] = lg_bcast(c, 0);
    v[1] = lg_bcast(c, 8);
    v[2] = lg_bcast(c, 16);
    v[3] = lg_bcast(c, 24);
}
// "the previous iterate is allclose to 0" (max |prev| <= 1e-8, evaluation.py:21-23) from the sum of squares
// wherever that decides (max^2 <= sum <= m max^2 over m entries); the max reduction only in between
template <typename F>
__device__ __forceinline__ bool lg_prev_is_zero(double sum2, double m, F &&max_abs) {
    if (sum2 > m * 1e-16) return false;
    if (sum2 <= 1e-16) return true;
    return max_abs() <= 1e-8;
}
// the rule with the two "previous iterate ~ 0" flags already known
__device__ __forceinline__ bool lg_rule_met2(double dx2, double x2, bool xs, double dz2, double z2, bool zs, double tol) {
    const double t2 = tol * tol * (1 + 1e-12);
    if (dx2 > t2 * (xs ? 1.0 : x2) || dz2 > t2 * (zs ? 1.0 : z2)) return false;
    const double rx = xs ? sqrt(dx2) : sqrt(dx2) / sqrt(x2);
    const double rz = zs ? sqrt(dz2) : sqrt(dz2) / sqrt(z2);
    return rx < tol && rz < tol;
}

// block-wide sum / max over VT threads (deterministic order); result in every thread
__device__ double block_sum(double v, double *sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wsum(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
    return t;
}
__device__ double block_max(double v, double *sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wmax(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = sh[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) t = fmax(t, sh[k]);
    return t;
}

struct LgDev {
    double *K, *Th, *C, *G, *W, *ThPrev;  // n x n
    const double *mask;                   // n x n or null (full)
    double *p, *x, *z, *uu, *v, *y;       // QP vectors (n); v and y are the ping-pong right-hand sides
    double *part;                         // ADMM norm partials [2][nblk][6]
    double *act;                          // 1.0 where the entry takes part in the QP of node u
    double *beta, *t, *a, *h, *gcol;      // post-QP vectors (n)
    double *gcol_next;                    // fused node update (round 4): column u + 1 of the UPDATED G, written by k_lg_update_g2
    const double *KT;                     // ... and K transposed (column u of K as a contiguous row); null: k_lg_qp_setup sets the QP up
    double *scal;                         // scalars: [0] done, [1] qp iterations, [2..] scratch
    int64_t n;
    double rho, qp_tol;
};

// y = A x, one wave per row (rows contiguous); skipped once the QP has stopped (chk_done)
__global__ __launch_bounds__(256) void k_matvec(const double *__restrict__ A, const double *__restrict__ x,
                                                double *__restrict__ y, int64_t n, const double *done, int chk_done) {
    if (chk_done && done[0] != 0.0) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double *a = A + row * n;
    double acc = 0;
    if ((n & 1) == 0) {  // (rows start 16-byte aligned: two entries per load, two chains)
        double acc2 = 0;
        for (int64_t j = 2 * lane; j < n; j += 128) {
            const double2 av = *reinterpret_cast<const double2 *>(a + j), xv = *reinterpret_cast<const double2 *>(x + j);
            acc += av.x * xv.x;
            acc2 += av.y * xv.y;
        }
        acc += acc2;
    } else {
        for (int64_t j = lane; j < n; j += 64) acc += a[j] * x[j];
    }
    acc = wsum(acc);
    if (lane == 0) y[row] = acc;
}

// K = x / mean * norm_par + H,  Theta = diag(1 / diag K),  C = diag(diag K)   (lgmrf.py:229,156-164)
__global__ void k_lg_prepare(const double *__restrict__ x, LgDev D, double mean, double norm_par, double hdiag) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.n) return;
    const double h = (i == j ? hdiag : 0.0) - 1.0;  // (2 or 1) * eye - ones
    const double k = x[i * D.n + j] / mean * norm_par + h;
    D.K[i * D.n + j] = k;
    D.Th[i * D.n + j] = (i == j) ? 1.0 / k : 0.0;
    D.C[i * D.n + j] = (i == j) ? k : 0.0;
}

// QP of node u: active set, p = k_u / k_uu, ADMM start (x = z = u = 0, v = rho p)   (lgmrf.py:177-202,55)
__global__ __launch_bounds__(VT) void k_lg_qp_setup(LgDev D, int64_t u) {
    const double kuu = D.K[u * D.n + u];
    for (int64_t i = threadIdx.x; i < D.n; i += VT) {
        const bool on = i != u && (D.mask == nullptr || D.mask[i * D.n + u] == 1.0);
        const double pi = on ? D.K[i * D.n + u] / kuu : 0.0;
        D.act[i] = on ? 1.0 : 0.0;
        D.p[i] = pi;
        D.x[i] = 0.0;
        D.z[i] = 0.0;
        D.uu[i] = 0.0;
        D.v[i] = on ? (0.0 - 0.0) + D.rho * pi : 0.0;
        D.gcol[i] = D.G ? D.G[i * D.n + u] : 0.0;
    }
    if (threadIdx.x == 0) {
        D.scal[0] = 0.0;
        D.scal[1] = 0.0;
    }
}

// In-place Gauss-Jordan step k (no pivoting: the matrix is SPD).  After n steps A holds A^-1.
__global__ void k_gj_pivot(const double *A, int64_t n, int64_t k, double *pivrow, double *pivcol) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double piv = A[k * n + k];
    pivrow[j] = (j == k) ? 1.0 / piv : A[k * n + j] / piv;  // scaled pivot row
    pivcol[j] = A[j * n + k];                               // pivot column before it is overwritten
}
__global__ void k_gj_eliminate(double *A, int64_t n, int64_t k, const double *__restrict__ pivrow,
                               const double *__restrict__ pivcol) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (i == k) {
        A[i * n + j] = pivrow[j];
        return;
    }
    const double cur = (j == k) ? 0.0 : A[i * n + j];
    A[i * n + j] = cur - pivcol[i] * pivrow[j];
}

// fast path: the explicit inverse for node u from G = (Theta + rho I)^-1,
//   inv(I + rho M) = I - rho (G_mm - g g'/G_uu)      (g = G[:,u]; identity on row/column u)
__global__ void k_lg_build_winv(LgDev D, int64_t u) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = D.n;
    if (j >= n) return;
    double w = (i == j) ? 1.0 : 0.0;
    if (i != u && j != u) w -= D.rho * (D.G[i * n + j] - (D.G[i * n + u] * D.G[j * n + u]) / D.G[u * n + u]);
    D.W[i * n + j] = w;
}

// One ADMM step (lgmrf.py:57-69) in ONE launch: workgroup = 8 rows of the explicit inverse W
// (2 per wave):  x' = W v,  z' = max(x' + u, 0),  u += x' - z',  next right-hand side
// v' = (z' - u) + rho p -- all row-local, so they are fused behind the row's dot product.
// The two relative changes need grid-wide norms: every workgroup leaves 6 partials and the
// stopping rule of step it-1 is evaluated (redundantly, in fixed order) during step it -- deferred
// by one launch, which costs nothing because a step that finds the rule satisfied exits before
// touching z.  Loads that do not depend on each other are issued up front (latency-bound kernel).
constexpr int ADMM_ROWS = 8;

__global__ __launch_bounds__(256) void k_lg_admm(LgDev D, int it) {
    __shared__ double sp[4][6];
    __shared__ int s_stop;
    if (D.scal[0] != 0.0) return;
    const int64_t n = D.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x;
    const double *__restrict__ vin = (it & 1) ? D.y : D.v;
    double *__restrict__ vout = (it & 1) ? D.v : D.y;
    const bool chk = it >= 2;  // rule of step it-1 (never for step 0: `it > 0`)

    // (1) loads for the deferred stopping rule (wave 0)
    double a0 = 0, a1 = 0, a3 = 0, a4 = 0, m2 = 0, m5 = 0;
    if (chk && wave == 0) {
        const double *pp = D.part + (size_t)((it - 1) & 1) * nblk * 6;
        for (int b = lane; b < nblk; b += 64) {
            a0 += pp[6 * b + 0];
            a1 += pp[6 * b + 1];
            m2 = fmax(m2, pp[6 * b + 2]);
            a3 += pp[6 * b + 3];
            a4 += pp[6 * b + 4];
            m5 = fmax(m5, pp[6 * b + 5]);
        }
    }
    // (2) my two rows: state of the row (lane r of the wave keeps row r's scalars) and the dot products
    const int64_t i0 = (int64_t)blockIdx.x * ADMM_ROWS + wave * 2;
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0;
    if (lane < 2 && i0 + lane < n) {
        const int64_t i = i0 + lane;
        xo = D.x[i];
        zo = D.z[i];
        uo = D.uu[i];
        pi = D.p[i];
        on = D.act[i];
    }
    double acc0 = 0, acc1 = 0;
    {
        const double *r0 = D.W + i0 * n, *r1 = r0 + n;
        const bool has0 = i0 < n, has1 = i0 + 1 < n;
        if ((n & 1) == 0) {  // 16-byte loads
            const double2 *v2 = reinterpret_cast<const double2 *>(vin);
            const double2 *q0 = reinterpret_cast<const double2 *>(r0), *q1 = reinterpret_cast<const double2 *>(r1);
#pragma unroll 4
            for (int64_t j = lane; j < n / 2; j += 64) {
                const double2 vv = v2[j];
                if (has0) {
                    const double2 a = q0[j];
                    acc0 += a.x * vv.x + a.y * vv.y;
                }
                if (has1) {
                    const double2 a = q1[j];
                    acc1 += a.x * vv.x + a.y * vv.y;
                }
            }
        } else {
            for (int64_t j = lane; j < n; j += 64) {
                const double vv = vin[j];
                if (has0) acc0 += r0[j] * vv;
                if (has1) acc1 += r1[j] * vv;
            }
        }
    }
    acc0 = wsum(acc0);
    acc1 = wsum(acc1);

    if (tid == 0) s_stop = 0;
    __syncthreads();
    if (chk && wave == 0) {
        a0 = wsum(a0);
        a1 = wsum(a1);
        a3 = wsum(a3);
        a4 = wsum(a4);
        m2 = wmax(m2);
        m5 = wmax(m5);
        if (lane == 0 && lg_rule_met(a0, a1, m2, a3, a4, m5, D.qp_tol)) {
            s_stop = 1;
            if (blockIdx.x == 0) {
                D.scal[1] = (double)it;  // steps run
                D.scal[0] = 1.0;
            }
        }
    }
    if (chk) __syncthreads();
    if (s_stop) return;

    double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
    if (lane < 2 && i0 + lane < n) {
        const int64_t i = i0 + lane;
        const bool active = on != 0.0;
        const double xn = active ? (lane == 0 ? acc0 : acc1) : 0.0;
        double zn = xn + uo;
        zn = zn < 0 ? 0.0 : zn;
        const double un = uo + (xn - zn);
        if (active) {
            dx2 = (xn - xo) * (xn - xo);
            x2 = xo * xo;
            xm = fabs(xo);
            dz2 = (zn - zo) * (zn - zo);
            z2 = zo * zo;
            zm = fabs(zo);
        }
        D.x[i] = xn;
        D.z[i] = active ? zn : 0.0;
        D.uu[i] = active ? un : 0.0;
        vout[i] = active ? (zn - un) + D.rho * pi : 0.0;
    }
    // lanes 0 and 1 of each wave hold partials: combine in fixed order (lane 0 + lane 1, then waves)
    dx2 += __shfl_xor(dx2, 1, 64);
    x2 += __shfl_xor(x2, 1, 64);
    xm = fmax(xm, __shfl_xor(xm, 1, 64));
    dz2 += __shfl_xor(dz2, 1, 64);
    z2 += __shfl_xor(z2, 1, 64);
    zm = fmax(zm, __shfl_xor(zm, 1, 64));
    if (lane == 0) {
        sp[wave][0] = dx2;
        sp[wave][1] = x2;
        sp[wave][2] = xm;
        sp[wave][3] = dz2;
        sp[wave][4] = z2;
        sp[wave][5] = zm;
    }
    __syncthreads();
    if (tid < 6) {
        double v;
        if (tid == 2 || tid == 5)
            v = fmax(fmax(sp[0][tid], sp[1][tid]), fmax(sp[2][tid], sp[3][tid]));
        else
            v = ((sp[0][tid] + sp[1][tid]) + sp[2][tid]) + sp[3][tid];
        D.part[((size_t)(it & 1) * nblk + blockIdx.x) * 6 + tid] = v;
    }
    if (blockIdx.x == 0 && tid == 0) D.scal[1] = (double)(it + 1);
}

// ------------------------------------------------------------------------------------------
// The same ADMM, all steps in ONE persistent launch ("resident" form): workgroup b keeps its R rows
// of the explicit inverse W in LDS for the whole QP (n = 2048: 8 rows = 128 KiB per CU, the 33.5 MB
// matrix is read from memory once per QP instead of once per step), the right-hand side v (n
// doubles) is exchanged through memory with write-through stores / sc1 loads, and the grid-wide
// dependency between steps is the XCD-hierarchical grid barrier of gl_gridbar.h instead of a
// kernel boundary.  Same update formulas, same deferred stopping rule (every workgroup evaluates
// it from the same partials in the same order).  Needs all workgroups co-resident (grid <= #CU,
// one workgroup per CU); the start-up rendezvous fails cleanly otherwise (error word in scal[8]).
constexpr int ADMM_RES_MAXR = 8;
#ifndef LG_HIST_LDS
#define LG_HIST_LDS 1    // tagged ADMM: the state one / two steps earlier in LDS (0: in registers, as before round 4)
#endif
#ifndef LG_RULE_EARLY
#define LG_RULE_EARLY 1  // ... the stopping rule of three steps ago is evaluated while the records are on their way (0: after the poll)
#endif
#ifndef LG_PROBES
#define LG_PROBES 1      // ... sweeps of the records a workgroup keeps in flight (2: a second one half a round trip behind)
#endif
#ifndef LG_PROBE_GAP
#define LG_PROBE_GAP 8
#endif
#ifndef LG_SELECTIVE
#define LG_SELECTIVE 1   // ... a retry re-reads only the records that are still missing (0: the whole sweep)
#endif

__device__ __forceinline__ double lg_ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void lg_st_sc1(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_lg_admm_res(LgDev D, GlGridBar gb, int R, int max_iter) {
    extern __shared__ double s_W[];  // R rows of W, row-major (R * n doubles)
    __shared__ double s_acc[4][ADMM_RES_MAXR];
    __shared__ double s_x[ADMM_RES_MAXR];
    __shared__ int s_ok, s_stop;
    if (D.scal[8] != 0.0) return;  // an earlier QP could not run (workgroups not co-resident): the host redoes the cycle
    const int64_t n = D.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nrows = (int)max((int64_t)0, min((int64_t)R, n - r0));

    for (int64_t q = tid; q < (int64_t)nrows * n; q += 256) s_W[q] = D.W[r0 * n + q];
    GlBarState bs{};
    if (tid == 0) s_ok = gl_barrier_setup(gb, bs) ? 1 : 0;
    __syncthreads();
    if (!s_ok) {
        if (blockIdx.x == 0 && tid == 0) D.scal[8] = 1.0;
        return;
    }
    // thread r < nrows owns row r0 + r: its ADMM state stays in registers for the whole QP
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0;
    if (tid < nrows) {
        const int64_t i = r0 + tid;
        xo = D.x[i];
        zo = D.z[i];
        uo = D.uu[i];
        pi = D.p[i];
        on = D.act[i];
    }
    int steps = 0;
    bool stopped = false;
    for (int it = 0; it < max_iter; ++it) {
        const double *vin = (it & 1) ? D.y : D.v;
        double *vout = (it & 1) ? D.v : D.y;
        const bool chk = it >= 2;  // rule of step it-1 (never for step 0: `it > 0`)
        // (1) partials of step it-1 for the deferred stopping rule (wave 0), requested first
        double a0 = 0, a1 = 0, a3 = 0, a4 = 0, m2 = 0, m5 = 0;
        if (chk && wave == 0) {
            const double *pp = D.part + (size_t)((it - 1) & 1) * nblk * 6;
            for (int b = lane; b < nblk; b += 64) {
                a0 += lg_ld_sc1(pp + 6 * b + 0);
                a1 += lg_ld_sc1(pp + 6 * b + 1);
                m2 = fmax(m2, lg_ld_sc1(pp + 6 * b + 2));
                a3 += lg_ld_sc1(pp + 6 * b + 3);
                a4 += lg_ld_sc1(pp + 6 * b + 4);
                m5 = fmax(m5, lg_ld_sc1(pp + 6 * b + 5));
            }
        }
        // (2) my rows times v: thread t takes columns t, t + 256, ...
        double acc[ADMM_RES_MAXR];
#pragma unroll
        for (int r = 0; r < ADMM_RES_MAXR; ++r) acc[r] = 0.0;
#pragma unroll 8
        for (int64_t j = tid; j < n; j += 256) {
            const double vj = lg_ld_sc1(vin + j);
#pragma unroll
            for (int r = 0; r < ADMM_RES_MAXR; ++r)
                if (r < nrows) acc[r] += s_W[(int64_t)r * n + j] * vj;
        }
#pragma unroll
        for (int r = 0; r < ADMM_RES_MAXR; ++r) acc[r] = wsum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < ADMM_RES_MAXR; ++r) s_acc[wave][r] = acc[r];
        }
        if (tid == 0) s_stop = 0;
        __syncthreads();
        if (tid < ADMM_RES_MAXR) s_x[tid] = ((s_acc[0][tid] + s_acc[1][tid]) + s_acc[2][tid]) + s_acc[3][tid];
        if (chk && wave == 0) {
            a0 = wsum(a0);
            a1 = wsum(a1);
            a3 = wsum(a3);
            a4 = wsum(a4);
            m2 = wmax(m2);
            m5 = wmax(m5);
            if (lane == 0 && lg_rule_met(a0, a1, m2, a3, a4, m5, D.qp_tol)) s_stop = 1;
        }
        __syncthreads();
        if (s_stop) {  // step it-1 met the rule: its z is the result, nothing of step `it` is applied
            stopped = true;
            break;
        }
        // (3) row-local updates (lgmrf.py:57-69) and the partial norms of this step
        double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
        if (tid < nrows) {
            const bool active = on != 0.0;
            const double xn = active ? s_x[tid] : 0.0;
            double zn = xn + uo;
            zn = zn < 0 ? 0.0 : zn;
            const double un = uo + (xn - zn);
            if (active) {
                dx2 = (xn - xo) * (xn - xo);
                x2 = xo * xo;
                xm = fabs(xo);
                dz2 = (zn - zo) * (zn - zo);
                z2 = zo * zo;
                zm = fabs(zo);
            }
            xo = xn;
            zo = active ? zn : 0.0;
            uo = active ? un : 0.0;
            lg_st_sc1(vout + r0 + tid, active ? (zn - un) + D.rho * pi : 0.0);
        }
        if (wave == 0) {  // rows live in lanes 0..R-1 of wave 0: fixed-order sums over those lanes
            dx2 = wsum(dx2);
            x2 = wsum(x2);
            xm = wmax(xm);
            dz2 = wsum(dz2);
            z2 = wsum(z2);
            zm = wmax(zm);
            if (lane == 0) {
                double *o = D.part + ((size_t)(it & 1) * nblk + blockIdx.x) * 6;
                lg_st_sc1(o + 0, dx2);
                lg_st_sc1(o + 1, x2);
                lg_st_sc1(o + 2, xm);
                lg_st_sc1(o + 3, dz2);
                lg_st_sc1(o + 4, z2);
                lg_st_sc1(o + 5, zm);
            }
        }
        steps = it + 1;
        if (!gl_grid_barrier(gb, bs, &s_ok)) {
            if (blockIdx.x == 0 && tid == 0) D.scal[8] = 2.0;  // barrier lost in mid-run
            break;
        }
    }
    if (tid < nrows) {
        const int64_t i = r0 + tid;
        D.x[i] = xo;
        D.z[i] = zo;
        D.uu[i] = uo;
    }
    if (blockIdx.x == 0 && tid == 0) {
        D.scal[1] = (double)steps;
        D.scal[0] = stopped ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// The persistent ADMM without a grid barrier ("tagged" form).  The only thing a step needs from the other
// workgroups is the right-hand side v, so v itself carries the synchronisation: entry j is published as
// a 16-byte record of two self-validating 8-byte granules {low word, tag}, {high word, tag} (one
// write-through store), tag = step the value is the input of, and a consumer simply re-reads its
// columns (sc1 loads, served past the L1) until every tag matches -- the hand-off recipe "R2: the data is
// the flag" of the CDNA4 guide.  One memory round trip per step instead of store + barrier + load, and
// one workgroup barrier.
//   * ping-pong of the record array by step parity is enough: a workgroup publishes tag s+1 only after it
//     has read every record of tag s, so nobody still reads the records (tag s-1) it overwrites;
//   * the stopping rule (lgmrf.py:66-69) needs all workgroups' norm partials of a step.  They are stored
//     (sc1) after the records of that step, and the storing wave drains (s_waitcnt vmcnt(0), free by then)
//     before it publishes the next records: whoever has seen a workgroup's tag s+2 may read its partials
//     of step s.  They are requested at the end of step s+2 and summed at step s+3 (their latency, 1.8 us
//     when waited for, hides behind a whole step), by every workgroup from the same values in the same
//     order, and three generations of the row state stay in registers to return the one of step s.
//     Partials live in a ring of four slots: a writer is never more than three steps ahead of a reader.
// Measured and rejected (round 2): a two-level hand-off -- one workgroup per XCD (elected per launch, XCD from
// HW_REG_XCC_ID) polls the records and republishes them with plain stores into a per-XCD copy that stays in that XCD's
// L2, the others poll the copy -- cuts the fabric traffic of a poll round from 8 MB to 256 KB at n = 2048 and is
// correct, but the extra hop costs more than the traffic did: 0.747 against 0.679 ms per node update at n = 2048,
// 0.504 against 0.455 at n = 1024.  A step is bound by hand-off latency, not by the bytes polled.
// Needs all workgroups co-resident like the barrier form; there is no rendezvous: the first poll is one (a
// workgroup that is not on the chip never publishes tag 0), every spin is bounded by the wall clock, and a
// run that gives up has written nothing but records.
typedef unsigned int lg_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lg_pack(unsigned lo, unsigned hi) {
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// NT = 256, RREG = 0: up to 8 rows per workgroup, all in LDS.  NT = 512, RREG = 8 ("wide", n > 1024): up to 16 rows, the
// first 8 in LDS and the others in the registers of the threads that multiply them (thread t: columns t, t + 512, ...) --
// half as many workgroups poll the n records of a step and half as many publish: at n = 2048 a poll round moves 4 MB
// through the fabric instead of 8.
// RLDS_ rows of a workgroup live in LDS, RREG in the registers of the threads that multiply them (their columns of them).
// <256, 0, 8>: up to 8 rows, all in LDS; <512, 8, 8> ("wide"): 16 rows, half and half; <512, 16, 0> ("regs", round 4): 16 rows,
// all in registers, four columns per thread (1024 or 768 threads leave 128 / 168 registers per wave: 45 / 21 of them spilled) -- the multiply reads no LDS at all (128 KB per step and workgroup in the wide form:
// 0.4 us of a 4.2 us step at n = 2048).
template <int NT, int RREG, int RLDS_ = ADMM_RES_MAXR>
__global__ __launch_bounds__(NT) void k_lg_admm_tag(LgDev D, unsigned *err, lg_u4 *gran, unsigned tag0, int R, int max_iter,
                                                    int64_t u_fast, int test_missing) {
    constexpr int CPT = (ADMM_RES_MAXR * 256 + NT - 1) / NT;  // columns per thread (n <= 2048)
    constexpr int NWV = NT / 64, RLDS = RLDS_, RTOT = RLDS_ + RREG;
    extern __shared__ double s_W[];  // the first min(R, 8) rows of W, row-major
    __shared__ double s_acc[2][NWV][RTOT];
    __shared__ int s_fail, s_stop[2];
    if (D.scal[8] != 0.0) return;  // an earlier QP could not run (workgroups not co-resident): the host redoes the cycle
    const int64_t n = D.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nrows = (int)max((int64_t)0, min((int64_t)R, n - r0));

    if (u_fast >= 0) {
        // fast path: my rows of the explicit inverse straight from G = (Theta + rho I)^-1, the formula of
        // k_lg_build_winv:  inv(I + rho M) = I - rho (G_mm - g g'/G_uu),  g = G[:,u] (left in gcol by
        // k_lg_qp_setup), identity on row / column u -- W itself is never written or read
        const double guu = D.G[u_fast * n + u_fast];
        for (int r = 0; r < min(nrows, RLDS); ++r) {
            const int64_t i = r0 + r;
            const double gq = D.gcol[i] / guu;  // (one quotient per row: a division per entry cost this prologue 5 us per launch)
            const double *grow = D.G + i * n;
            for (int j = tid; j < (int)n; j += NT) {
                double w = (i == j) ? 1.0 : 0.0;
                if (i != u_fast && j != u_fast) w -= D.rho * (grow[j] - gq * D.gcol[j]);
                s_W[r * (int)n + j] = w;
            }
        }
    } else {
        for (int64_t q = tid; q < (int64_t)min(nrows, RLDS) * n; q += NT) s_W[q] = D.W[r0 * n + q];
    }
    double wr[RREG > 0 ? RREG : 1][CPT];  // rows 8 .. 15 of my workgroup: my columns of them
    if (RREG > 0) {
        const double guu = u_fast >= 0 ? D.G[u_fast * n + u_fast] : 1.0;
#pragma unroll
        for (int r = 0; r < RREG; ++r) {
            const int64_t i = r0 + RLDS + r;
            const bool row = RLDS + r < nrows;
            const double gq = (row && u_fast >= 0) ? D.gcol[i] / guu : 0.0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int64_t j = tid + NT * (int64_t)c;
                double w = 0.0;
                if (row && j < n) {
                    if (u_fast >= 0) {
                        w = (i == j) ? 1.0 : 0.0;
                        if (i != u_fast && j != u_fast) w -= D.rho * (D.G[i * n + j] - gq * D.gcol[j]);
                    } else {
                        w = D.W[i * n + j];
                    }
                }
                wr[r][c] = w;
            }
        }
    }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gran, 0, (int)(2 * n * 16), 0x00020000);
    const auto prsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D.part, 0, (int)(4 * nblk * 48), 0x00020000);
    auto publish = [&](int parity, int64_t i, double v, unsigned tag) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        lg_u4 g;
        g.x = (unsigned)b, g.y = tag, g.z = (unsigned)(b >> 32), g.w = tag;
        __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (int)((parity * n + i) * 16), 0, 16);  // aux 16 = sc1
    };
    // thread r < nrows owns row r0 + r; three generations of its ADMM state stay in registers
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0;
    // the state one / two steps earlier: in LDS, [generation][x, z, u][row] -- only the threads that own a row need it, and the
    // sixteen registers it took in every thread put the wide form into scratch
    __shared__ double s_hist[2][3][RTOT];
    if (tid < RTOT) {
#pragma unroll
        for (int g = 0; g < 2; ++g) s_hist[g][0][tid] = s_hist[g][1][tid] = s_hist[g][2][tid] = 0.0;
    }
#if !LG_HIST_LDS
    double x1 = 0, z1 = 0, u1 = 0, x2g = 0, z2g = 0, u2g = 0;
#endif
    if (tid < nrows) {
        const int64_t i = r0 + tid;
        double v0;
        if (D.KT != nullptr && u_fast >= 0) {
            // the QP's set-up (k_lg_qp_setup: lgmrf.py:177-202,55) for my rows, here: active set = everybody but u (full mask),
            // p = k_u / k_uu from the transposed K (column u as a contiguous row), x = z = u = 0, v = rho p
            on = (i != u_fast) ? 1.0 : 0.0;
            pi = on != 0.0 ? D.KT[u_fast * n + i] / D.K[u_fast * n + u_fast] : 0.0;
            v0 = on != 0.0 ? (0.0 - 0.0) + D.rho * pi : 0.0;
        } else {
            xo = D.x[i];
            zo = D.z[i];
            uo = D.uu[i];
            pi = D.p[i];
            on = D.act[i];
            v0 = D.v[i];
        }
        if (!((test_missing & 1) && blockIdx.x == 0)) publish(0, i, v0, tag0);  // the input of step 0
    }
    // (sixteen rows per workgroup: at most 2048 / 16 = 128 workgroups, two per lane)
    constexpr int NQ = RTOT >= 16 ? 2 : 4;
    lg_u4 pr[NQ][3];  // wave 0: the partials {dx2, x2 | xm, dz2 | z2, zm} of workgroups lane, lane + 64, ...
    bool pending = false;
    int steps = 0;
    bool stopped = false, failed = false;
    for (int it = 0; it <= max_iter + 1; ++it) {
        // (1) my columns of v: thread t takes columns t, t + NT, ...  All of a thread's records are requested at once; what does
        // not carry this step's tag yet is re-read -- only that: a lane whose record has arrived keeps it (before round 4
        // the whole sweep, n records per workgroup, was repeated until every tag matched).
        const unsigned tag = tag0 + (unsigned)it;
        lg_u4 gq[CPT];
        if (it <= max_iter) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int64_t j = tid + NT * (int64_t)c;
                gq[c] = lg_u4{0u, tag, 0u, tag};
                if (j < n) gq[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((it & 1) * n) + j) * 16), 0, 16);
            }
        }
        auto rule_of_three_steps_ago = [&]() {
        // (2) the rule of step it-3 from the partials requested one step ago -- wave 0, while the records above are on their
            // way (in front of the multiply it held the whole workgroup up at the barrier below)
            if (wave == 0) {
                int stop = 0;
                if (pending) {
                    double a0 = 0, a1 = 0, a3 = 0, a4 = 0, m2 = 0, m5 = 0;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (lane + 64 * q < nblk) {
                            a0 += lg_pack(pr[q][0].x, pr[q][0].y);
                            a1 += lg_pack(pr[q][0].z, pr[q][0].w);
                            m2 = fmax(m2, lg_pack(pr[q][1].x, pr[q][1].y));
                            a3 += lg_pack(pr[q][1].z, pr[q][1].w);
                            a4 += lg_pack(pr[q][2].x, pr[q][2].y);
                            m5 = fmax(m5, lg_pack(pr[q][2].z, pr[q][2].w));
                        }
                    a0 = wsum(a0);
                    a1 = wsum(a1);
                    a3 = wsum(a3);
                    a4 = wsum(a4);
                    m2 = wmax(m2);
                    m5 = wmax(m5);
                    stop = lg_rule_met(a0, a1, m2, a3, a4, m5, D.qp_tol) ? 1 : 0;
                }
                if (lane == 0) s_stop[it & 1] = stop;
            }
        };
#if LG_RULE_EARLY
        rule_of_three_steps_ago();
#endif
        double vj[CPT];
#if LG_PROBES == 2
        // a second sweep in flight, half a round trip behind the first: what a probe finds is decided when it reaches memory
        lg_u4 gq2[CPT];
        if (it <= max_iter) {
            __builtin_amdgcn_s_sleep(LG_PROBE_GAP);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int64_t j = tid + NT * (int64_t)c;
                gq2[c] = lg_u4{0u, tag, 0u, tag};
                if (j < n) gq2[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((it & 1) * n) + j) * 16), 0, 16);
            }
            asm volatile("" ::: "memory");
        }
#endif
        if (it <= max_iter) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool good = true;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < CPT; ++c) ok = ok && gq[c].y == tag && gq[c].w == tag;
                if (__all(ok)) break;
#if LG_PROBES == 2
                {   // the older sweep missed: request it again and look at the younger one meanwhile
                    lg_u4 t[CPT];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        const int64_t j = tid + NT * (int64_t)c;
                        t[c] = gq2[c];
                        gq2[c] = lg_u4{0u, tag, 0u, tag};
                        if (j < n) gq2[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((it & 1) * n) + j) * 16), 0, 16);
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < CPT; ++c) gq[c] = t[c];
                    if ((spins & 63u) == 63u &&
                        (gl_ld_u32(err) != 0u || __builtin_amdgcn_s_memrealtime() - t_start > 100000000ull)) {
                        good = false;
                        break;
                    }
                    continue;
                }
#endif
                // ~0.1 us: leave the fabric to the stores we wait for (development: GRAPHLEARN_AMD_LGMRF_POLL = 1 / 2 / 3)
                const int pm = test_missing >> 1;
                if (pm == 0) __builtin_amdgcn_s_sleep(4);
                else if (pm == 1) __builtin_amdgcn_s_sleep(1);
                else if (pm == 3) __builtin_amdgcn_s_sleep(12);
                if ((spins & 63u) == 63u &&
                    (gl_ld_u32(err) != 0u || __builtin_amdgcn_s_memrealtime() - t_start > 100000000ull)) {
                    good = false;  // 1 s: a producer is not on the chip
                    break;
                }
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int64_t j = tid + NT * (int64_t)c;
                    if (j < n && (!LG_SELECTIVE || !(gq[c].y == tag && gq[c].w == tag)))
                        gq[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((it & 1) * n) + j) * 16), 0, 16);
                }
                asm volatile("" ::: "memory");  // the loads stay in the loop
            }
            if (!good) s_fail = 1;
#pragma unroll
            for (int c = 0; c < CPT; ++c) vj[c] = lg_pack(gq[c].x, gq[c].z);
        }
#if !LG_RULE_EARLY
        rule_of_three_steps_ago();
#endif
        // (3) my rows times v
        if (it < max_iter) {
            if constexpr (RLDS > 0) {
                double acc[RLDS > 0 ? RLDS : 1];
#pragma unroll
                for (int r = 0; r < RLDS; ++r) acc[r] = 0.0;
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int64_t j = tid + NT * (int64_t)c;
                    if (j < n) {
#pragma unroll
                        for (int r = 0; r < RLDS; ++r)
                            if (r < nrows) acc[r] += s_W[(int64_t)r * n + j] * vj[c];
                    }
                }
                // the eight row sums of the wave in ONE halving butterfly (10 exchanges instead of 8 x 6): afterwards
                // lane 4 r holds row r
                const double rs = add_xor32(halfwave_rows8(acc, lane));
                if (lane < 32 && (lane & 3) == 0) s_acc[it & 1][wave][lane >> 2] = rs;
            }
#pragma unroll
            for (int gr = 0; gr < RREG / 8; ++gr) {  // the rows kept in registers (zero where there is no row / column: wr is)
                double acr[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acr[r] = 0.0;
#pragma unroll
                    for (int c = 0; c < CPT; ++c) acr[r] += wr[RREG > 0 ? 8 * gr + r : 0][c] * vj[c];
                }
                const double rs2 = add_xor32(halfwave_rows8(acr, lane));
                if (lane < 32 && (lane & 3) == 0) s_acc[it & 1][wave][RLDS + 8 * gr + (lane >> 2)] = rs2;
            }
        }
        __syncthreads();
        if (s_fail) {
            failed = true;
            steps = it;
            break;
        }
        if (s_stop[it & 1]) {  // step it-3 met the rule: its z is the result, the steps after it are dropped
            stopped = true;
#if LG_HIST_LDS
            if (tid < nrows) {
                const int g = it <= max_iter ? 1 : 0;  // (the last pass follows a pass without an update)
                xo = s_hist[g][0][tid], zo = s_hist[g][1][tid], uo = s_hist[g][2][tid];
            }
#else
            if (it <= max_iter) xo = x2g, zo = z2g, uo = u2g;
            else xo = x1, zo = z1, uo = u1;
#endif
            steps = it - 2;
            break;
        }
        // (4) row-local updates (lgmrf.py:57-69), the records of the next step, the partial norms of this one
        if (it < max_iter) {
            double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
            if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my partials of step it-1 have left before tag it+1 does
            if (tid < nrows) {
                const bool active = on != 0.0;
                double sx = ((s_acc[it & 1][0][tid] + s_acc[it & 1][1][tid]) + s_acc[it & 1][2][tid]) + s_acc[it & 1][3][tid];
#pragma unroll
                for (int v = 4; v < NWV; ++v) sx += s_acc[it & 1][v][tid];
                const double xn = active ? sx : 0.0;
                double zn = xn + uo;
                zn = zn < 0 ? 0.0 : zn;
                const double un = uo + (xn - zn);
                if (active) {
                    dx2 = (xn - xo) * (xn - xo);
                    x2 = xo * xo;
                    xm = fabs(xo);
                    dz2 = (zn - zo) * (zn - zo);
                    z2 = zo * zo;
                    zm = fabs(zo);
                }
#if LG_HIST_LDS
                s_hist[1][0][tid] = s_hist[0][0][tid], s_hist[1][1][tid] = s_hist[0][1][tid], s_hist[1][2][tid] = s_hist[0][2][tid];
                s_hist[0][0][tid] = xo, s_hist[0][1][tid] = zo, s_hist[0][2][tid] = uo;
#else
                x2g = x1, z2g = z1, u2g = u1;
                x1 = xo, z1 = zo, u1 = uo;
#endif
                xo = xn;
                zo = active ? zn : 0.0;
                uo = active ? un : 0.0;
                publish((it + 1) & 1, r0 + tid, active ? (zn - un) + D.rho * pi : 0.0, tag + 1u);
            }
            if (wave == 0) {  // rows live in lanes 0..R-1 of wave 0: fixed-order sums over those lanes
                double sq[4] = {dx2, x2, dz2, z2};
                lg_wsum4(sq, lane);
                dx2 = sq[0], x2 = sq[1], dz2 = sq[2], z2 = sq[3];
                xm = wmax(xm);
                zm = wmax(zm);
                if (lane == 0) {
                    const int off = ((it & 3) * nblk + (int)blockIdx.x) * 48;
                    auto st2 = [&](int o, double a, double b) {
                        const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
                        lg_u4 g;
                        g.x = (unsigned)ua, g.y = (unsigned)(ua >> 32), g.z = (unsigned)ub, g.w = (unsigned)(ub >> 32);
                        __builtin_amdgcn_raw_buffer_store_b128(g, prsrc, off + o, 0, 16);
                    };
                    st2(0, dx2, x2);
                    st2(16, xm, dz2);
                    st2(32, z2, zm);
                }
            }
            steps = it + 1;
        }
        // (5) request the partials of step it-2 (every workgroup's tag `it` has been seen); summed one step later
        pending = false;
        if (it <= max_iter && it >= 3) {  // never for step 0 (`it > 0` in the reference's rule)
            pending = true;
            if (wave == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (lane + 64 * q < nblk) {
                        const int off = (((it - 2) & 3) * nblk + lane + 64 * q) * 48;
                        pr[q][0] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, off, 0, 16);
                        pr[q][1] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, off + 16, 0, 16);
                        pr[q][2] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, off + 32, 0, 16);
                    }
            }
        }
    }
    if (failed) {  // nothing of the QP's state has been written: the rows live in registers until the end
        if (tid == 0) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // at step 0: the workgroups are not all on the chip (the host falls back to one launch per step);
            // later: a hand-off was lost in mid-run
            D.scal[8] = steps == 0 ? 1.0 : 2.0;
        }
        return;
    }
    if (tid < nrows) {
        const int64_t i = r0 + tid;
        D.x[i] = xo;
        D.z[i] = zo;
        D.uu[i] = uo;
        if (u_fast >= 0) {  // fast path of the fit: k_lg_beta and k_lg_snap_col ride along
            D.beta[i] = (on != 0.0) ? -zo : 0.0;
            D.h[i] = D.C[i * n + u_fast];
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        D.scal[1] = (double)steps;
        D.scal[0] = stopped ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// The ADMM of small problems (n <= 64) in ONE WAVE: lane i keeps row i of the explicit inverse in
// registers, the right-hand side lives one entry per lane and is broadcast entry by entry through the
// scalar file (v_readlane), the norms are DPP reductions, the stopping rule (lgmrf.py:66-69) is evaluated
// in place after every step -- no LDS, no barrier, no memory traffic inside the loop: ~0.5 us per step
// against 1.7 us for the multi-workgroup forms at this size.  NP = n rounded up to 16 / 32 / 64 (zero
// padding).  u_fast >= 0: rows of inv(I + rho M) straight from G (the formula of k_lg_build_winv).

template <int NP>
__global__ __launch_bounds__(64) void k_lg_admm_wave(LgDev D, int max_iter, int64_t u_fast) {
    const int n = (int)D.n, lane = threadIdx.x;
    const bool rowon = lane < n;
    double w[NP];
    if (u_fast >= 0) {
        const int u = (int)u_fast;
        const double guu = D.G[(int64_t)u * n + u];
        const double gi = rowon ? D.gcol[lane] : 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            double t = 0.0;
            if (rowon && j < n) {
                t = (lane == j) ? 1.0 : 0.0;
                if (lane != u && j != u) t -= D.rho * (D.G[(int64_t)lane * n + j] - (gi * D.gcol[j]) / guu);
            }
            w[j] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) w[j] = (rowon && j < n) ? D.W[(int64_t)lane * n + j] : 0.0;
    }
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0, v = 0;
    if (rowon) {
        xo = D.x[lane];
        zo = D.z[lane];
        uo = D.uu[lane];
        pi = D.p[lane];
        on = D.act[lane];
        v = D.v[lane];
    }
    const bool active = on != 0.0;
    int steps = 0;
    bool stopped = false;
    for (int it = 0; it < max_iter; ++it) {
        // four independent fused chains: a lone wave is bound by the latency of a dependent fp64 add per column
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            a0 = __builtin_fma(w[j], lg_bcast(v, j), a0);
            a1 = __builtin_fma(w[j + 1], lg_bcast(v, j + 1), a1);
            a2 = __builtin_fma(w[j + 2], lg_bcast(v, j + 2), a2);
            a3 = __builtin_fma(w[j + 3], lg_bcast(v, j + 3), a3);
        }
        const double acc = (a0 + a1) + (a2 + a3);
        const double xn = active ? acc : 0.0;
        double zn = xn + uo;
        zn = zn < 0 ? 0.0 : zn;
        const double un = uo + (xn - zn);
        double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
        if (active) {
            dx2 = (xn - xo) * (xn - xo);
            x2 = xo * xo;
            xm = fabs(xo);
            dz2 = (zn - zo) * (zn - zo);
            z2 = zo * zo;
            zm = fabs(zo);
        }
        xo = xn;
        zo = active ? zn : 0.0;
        uo = active ? un : 0.0;
        v = active ? (zn - un) + D.rho * pi : 0.0;
        steps = it + 1;
        if (it > 0) {  // `if it > 0 and ...`; relative_error's plain-norm branch when prev ~ 0
            double sq[4] = {dx2, x2, dz2, z2};
            lg_wsum4(sq, lane);
            const bool xs = lg_prev_is_zero(sq[1], 64.0, [&] { return wmax(xm); });
            const bool zs = lg_prev_is_zero(sq[3], 64.0, [&] { return wmax(zm); });
            if (lg_rule_met2(sq[0], sq[1], xs, sq[2], sq[3], zs, D.qp_tol)) {
                stopped = true;
                break;
            }
        }
    }
    if (rowon) {
        D.x[lane] = xo;
        D.z[lane] = zo;
        D.uu[lane] = uo;
        if (u_fast >= 0) {  // fast path of the fit: k_lg_beta and k_lg_snap_col ride along
            D.beta[lane] = active ? -zo : 0.0;
            D.h[lane] = D.C[(int64_t)lane * n + u_fast];
        }
    }
    if (lane == 0) {
        D.scal[1] = (double)steps;
        D.scal[0] = stopped ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// The ADMM of mid-size problems (64 < n <= 160) in ONE workgroup with the explicit inverse in REGISTERS:
// four lanes per row, lane h of row r keeps the entries W[r][4c + h], c < CPL (CPL = 32: n <= 128, 512
// threads; CPL = 40: n <= 160, 640 threads -- 80 of the 170 registers a thread has there; 48 spills), the right-hand
// side sits in LDS (double-buffered, broadcast reads), the four partial dot products of a row meet by two
// DPP adds, lane 0 of the quad updates the row.  One workgroup barrier per step (norm partials and the next
// right-hand side are double-buffered), no traffic outside the CU: ~0.6-0.9 us per step against 3.3 us for
// the multi-workgroup form at these sizes.  u_fast >= 0: rows of inv(I + rho M) straight from G.
constexpr int ADMM_QUAD_MAXN = 160;

template <int CPL>
__global__ __launch_bounds__(16 * CPL) void k_lg_admm_quad(LgDev D, int max_iter, int64_t u_fast) {
    constexpr int NR = 4 * CPL, NT = 4 * NR, NW = NT / 64;
    __shared__ double s_v[2][NR];
    __shared__ double s_red[2][6][NW];
    const int n = (int)D.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = tid >> 2, h = tid & 3;
    const bool rowon = row < n;
    double w[CPL];
    if (u_fast >= 0) {
        const int u = (int)u_fast;
        const double guu = D.G[(int64_t)u * n + u];
        const double gi = rowon ? D.gcol[row] : 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = 4 * c + h;
            double t = 0.0;
            if (rowon && j < n) {
                t = (row == j) ? 1.0 : 0.0;
                if (row != u && j != u) t -= D.rho * (D.G[(int64_t)row * n + j] - (gi * D.gcol[j]) / guu);
            }
            w[c] = t;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = 4 * c + h;
            w[c] = (rowon && j < n) ? D.W[(int64_t)row * n + j] : 0.0;
        }
    }
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0;
    if (rowon && h == 0) {
        xo = D.x[row];
        zo = D.z[row];
        uo = D.uu[row];
        pi = D.p[row];
        on = D.act[row];
    }
    if (h == 0) s_v[0][row] = rowon ? D.v[row] : 0.0;
    const bool active = on != 0.0;  // false on lanes h != 0 and on padded rows
    __syncthreads();
    int steps = 0;
    bool stopped = false;
    for (int it = 0; it < max_iter; ++it) {
        const double *vin = s_v[it & 1];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four independent fused chains
#pragma unroll
        for (int c = 0; c < CPL; c += 4) {
            a0 = __builtin_fma(w[c], vin[4 * c + h], a0);
            a1 = __builtin_fma(w[c + 1], vin[4 * c + 4 + h], a1);
            a2 = __builtin_fma(w[c + 2], vin[4 * c + 8 + h], a2);
            a3 = __builtin_fma(w[c + 3], vin[4 * c + 12 + h], a3);
            if (CPL > 32 && (c & 7) == 4) __builtin_amdgcn_sched_barrier(0);  // 640 threads: eight LDS reads in flight are enough
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += dpp_mov<DPP_QUAD_XOR1>(acc);
        acc += dpp_mov<DPP_QUAD_XOR2>(acc);
        double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
        {
            const double xn = active ? acc : 0.0;
            double zn = xn + uo;
            zn = zn < 0 ? 0.0 : zn;
            const double un = uo + (xn - zn);
            if (active) {
                dx2 = (xn - xo) * (xn - xo);
                x2 = xo * xo;
                xm = fabs(xo);
                dz2 = (zn - zo) * (zn - zo);
                z2 = zo * zo;
                zm = fabs(zo);
            }
            xo = xn;
            zo = active ? zn : 0.0;
            uo = active ? un : 0.0;
            if (h == 0) s_v[(it + 1) & 1][row] = active ? (zn - un) + D.rho * pi : 0.0;
        }
        double sq[4] = {dx2, x2, dz2, z2};
        lg_wsum4(sq, lane);  // per-wave sums, all four in every lane
        if (lane == 0) {
            double *r = &s_red[it & 1][0][0];
            r[0 * NW + wave] = sq[0];
            r[1 * NW + wave] = sq[1];
            r[2 * NW + wave] = sq[2];
            r[3 * NW + wave] = sq[3];
        }
        __syncthreads();  // the next right-hand side and this step's norm partials are out
        steps = it + 1;
        if (it > 0) {  // `if it > 0 and ...`; relative_error's plain-norm branch when prev ~ 0
            const double *r = &s_red[it & 1][0][0];
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double a = r[q * NW];
#pragma unroll
                for (int v = 1; v < NW; ++v) a += r[q * NW + v];
                t[q] = a;
            }
            // max |prev| over the workgroup, only when the sums of squares do not already tell (same branch everywhere)
            auto block_max = [&](double m) {
                m = wmax(m);
                double *mx = &s_red[it & 1][4][0];
                if (lane == 0) mx[wave] = m;
                __syncthreads();
                double a = mx[0];
                for (int v = 1; v < NW; ++v) a = fmax(a, mx[v]);
                __syncthreads();
                return a;
            };
            const bool xs = lg_prev_is_zero(t[1], (double)n, [&] { return block_max(xm); });
            const bool zs = lg_prev_is_zero(t[3], (double)n, [&] { return block_max(zm); });
            if (lg_rule_met2(t[0], t[1], xs, t[2], t[3], zs, D.qp_tol)) {
                stopped = true;
                break;
            }
        }
    }
    if (rowon && h == 0) {
        D.x[row] = xo;
        D.z[row] = zo;
        D.uu[row] = uo;
        if (u_fast >= 0) {  // fast path of the fit: k_lg_beta and k_lg_snap_col ride along
            D.beta[row] = active ? -zo : 0.0;
            D.h[row] = D.C[(int64_t)row * n + u_fast];
        }
    }
    if (tid == 0) {
        D.scal[1] = (double)steps;
        D.scal[0] = stopped ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// The same ADMM for small problems (n <= 64) in ONE launch of ONE workgroup: the whole explicit inverse
// (<= 128 KiB) in LDS (transposed, so a wave's lanes read consecutive addresses), v double-buffered in
// LDS, 8 threads per row, row sums combined by DPP, the
// stopping rule evaluated in place after each step exactly as the reference does (lgmrf.py:66-69).
// A step costs ~1 us instead of a ~5 us launch.
constexpr int ADMM_ONE_MAXN = 64;  // measured: n = 32: 0.25 -> 0.21 ms per node-update, 64: 0.40 -> 0.35, 128: 0.61 -> 0.83 (slower)

__global__ __launch_bounds__(1024) void k_lg_admm_one(LgDev D, int max_iter) {
    extern __shared__ double s_W[];  // n * n, then v[2][n], then red[6][16]
    const int n = (int)D.n;
    double *s_v = s_W + (size_t)n * n;
    double *red = s_v + 2 * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = tid >> 3, part = tid & 7;
    // transposed copy: s_W[j * n + r] = W[r][j], so that the 8 x 8 (row, part) lanes of a wave read
    // consecutive addresses (no symmetry of W is assumed)
    for (int q = tid; q < n * n; q += 1024) s_W[(size_t)(q % n) * n + q / n] = D.W[q];
    if (tid < n) s_v[tid] = D.v[tid];
    double xo = 0, zo = 0, uo = 0, pi = 0, on = 0;
    if (part == 0 && row < n) {
        xo = D.x[row];
        zo = D.z[row];
        uo = D.uu[row];
        pi = D.p[row];
        on = D.act[row];
    }
    __syncthreads();
    int steps = 0;
    bool stopped = false;
    for (int it = 0; it < max_iter; ++it) {
        const double *vin = s_v + (it & 1) * n;
        double *vout = s_v + ((it + 1) & 1) * n;
        double acc = 0.0;
        if (row < n)
            for (int j = part; j < n; j += 8) acc += s_W[(size_t)j * n + row] * vin[j];
        acc += dpp_mov<DPP_QUAD_XOR1>(acc);
        acc += dpp_mov<DPP_QUAD_XOR2>(acc);
        acc += dpp_mov<DPP_HALF_MIRROR>(acc);
        double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
        if (part == 0 && row < n) {
            const bool active = on != 0.0;
            const double xn = active ? acc : 0.0;
            double zn = xn + uo;
            zn = zn < 0 ? 0.0 : zn;
            const double un = uo + (xn - zn);
            if (active) {
                dx2 = (xn - xo) * (xn - xo);
                x2 = xo * xo;
                xm = fabs(xo);
                dz2 = (zn - zo) * (zn - zo);
                z2 = zo * zo;
                zm = fabs(zo);
            }
            xo = xn;
            zo = active ? zn : 0.0;
            uo = active ? un : 0.0;
            vout[row] = active ? (zn - un) + D.rho * pi : 0.0;
        }
        dx2 = wsum(dx2);
        x2 = wsum(x2);
        xm = wmax(xm);
        dz2 = wsum(dz2);
        z2 = wsum(z2);
        zm = wmax(zm);
        if (lane == 0) {
            red[0 * 16 + wave] = dx2;
            red[1 * 16 + wave] = x2;
            red[2 * 16 + wave] = xm;
            red[3 * 16 + wave] = dz2;
            red[4 * 16 + wave] = z2;
            red[5 * 16 + wave] = zm;
        }
        __syncthreads();
        double t[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double a = (q == 2 || q == 5) ? 0.0 : 0.0;
#pragma unroll
            for (int v = 0; v < 16; ++v) a = (q == 2 || q == 5) ? fmax(a, red[q * 16 + v]) : a + red[q * 16 + v];
            t[q] = a;
        }
        steps = it + 1;
        if (it > 0) {  // lgmrf.py:68 `if it > 0 and ...`; relative_error's plain-norm branch when prev ~ 0
            if (lg_rule_met(t[0], t[1], t[2], t[3], t[4], t[5], D.qp_tol)) {
                stopped = true;
                break;
            }
        }
        __syncthreads();  // red and the v buffer about to be overwritten are free again
    }
    if (part == 0 && row < n) {
        D.x[row] = xo;
        D.z[row] = zo;
        D.uu[row] = uo;
    }
    if (tid == 0) {
        D.scal[1] = (double)steps;
        D.scal[0] = stopped ? 1.0 : 0.0;
    }
}

// The QP's ADMM loop: launches in chunks and polls the device-side stop flag in between, so a
// QP that stops after ~30 steps does not pay for 100 launches.
// Persistent form: rows per workgroup R = ceil(n / #CU) <= 8 and R * n * 8 bytes of LDS.
// ------------------------------------------------------------------------------------------
// The whole block-coordinate descent of a small problem (n <= 64) in ONE launch of ONE workgroup.  Fast path
// (full mask) as described next; GENERAL (any mask): no G, per node W = I + rho M on the active set is built
// from C in the same LDS matrix and inverted in place (k_lg_build_w + Gauss-Jordan of the launch path).  lgmrf.py:169-226: K', Theta, C = Theta^-1 and G = (Theta + rho I)^-1 live in LDS (row
// stride n + 1), a cycle inverts Theta + rho I by Gauss-Jordan in place, and per node: the QP's set-up, its
// ADMM in wave 0 (rows of inv(I + rho M) in registers, right-hand side broadcast through the scalar file,
// as k_lg_admm_wave), beta, C beta, the rank-one updates of Theta and C and the rank-two Woodbury
// correction of G -- the same element-wise formulas as the kernels of the launch path (k_lg_qp_setup,
// k_lg_build_winv, k_lg_node_update, k_lg_update_c, k_lg_wood_scal, k_lg_update_g), separated by workgroup
// barriers instead of eleven launches.  The cycle rule (:221-224) is evaluated in place.
// At n = 32 a node update costs ~50 us here against 96 us with eleven launches around a one-wave ADMM.
template <int NP, bool GENERAL>
__global__ __launch_bounds__(256) void k_lg_fit_small(LgDev D, int max_cycle, double prob_tol, int max_qp_iter, double *out) {
    extern __shared__ double sm[];
    __shared__ double s_red[3][4];
    __shared__ double s_sc[8];
    const int n = (int)D.n, S = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *sKT = sm, *sTh = sKT + n * S, *sC = sTh + n * S, *sG = sC + n * S;
    double *s_beta = sG + n * S, *s_t = s_beta + n, *s_a = s_t + n, *s_h = s_a + n, *s_g = s_h + n;
    double *s_prow = s_g + n, *s_pcol = s_prow + n;
    const bool col = lane < n;
    const double rho = D.rho;

    for (int i = wave; i < n; i += 4)
        if (col) {
            sKT[lane * S + i] = D.K[(int64_t)i * n + lane];  // transposed: column u of K is row u here
            sTh[i * S + lane] = D.Th[(int64_t)i * n + lane];
            sC[i * S + lane] = D.C[(int64_t)i * n + lane];
        }
    __syncthreads();

    int converged = -1, cycles = 0;
    for (int cycle = 0; cycle < max_cycle; ++cycle) {
        // l_pre (:173) to memory (only the cycle rule reads it back); G0 = Theta + rho I
        for (int i = wave; i < n; i += 4)
            if (col) {
                D.ThPrev[(int64_t)i * n + lane] = sTh[i * S + lane];
                if (!GENERAL) sG[i * S + lane] = sTh[i * S + lane] + (i == lane ? rho : 0.0);
            }
        __syncthreads();
        // Gauss-Jordan in place, no pivoting (SPD), as k_gj_pivot / k_gj_eliminate
        auto invert_sG = [&]() {
            for (int k = 0; k < n; ++k) {
                if (tid < n) {
                    const double piv = sG[k * S + k];
                    s_prow[tid] = (tid == k) ? 1.0 / piv : sG[k * S + tid] / piv;
                    s_pcol[tid] = sG[tid * S + k];
                }
                __syncthreads();
                for (int i = wave; i < n; i += 4)
                    if (col) {
                        if (i == k) {
                            sG[i * S + lane] = s_prow[lane];
                        } else {
                            const double cur = (lane == k) ? 0.0 : sG[i * S + lane];
                            sG[i * S + lane] = cur - s_pcol[i] * s_prow[lane];
                        }
                    }
                __syncthreads();
            }
        };
        if (!GENERAL) invert_sG();  // fast path: G = (Theta + rho I)^-1 once per cycle

        for (int u = 0; u < n; ++u) {
            if (GENERAL) {  // k_lg_build_w: W = I + rho M on the active set (identity elsewhere), M = C - c c'/c_uu; inverted in place
                if (tid < n) s_a[tid] = (tid != u && D.mask[(int64_t)tid * n + u] == 1.0) ? 1.0 : 0.0;  // the active set (s_a is free here)
                __syncthreads();
                const double cuu = sC[u * S + u];
                for (int i = wave; i < n; i += 4)
                    if (col) {
                        double t = (i == lane) ? 1.0 : 0.0;
                        if (s_a[i] != 0.0 && s_a[lane] != 0.0) {
                            const double m = sC[i * S + lane] - (sC[i * S + u] * sC[lane * S + u]) / cuu;
                            t += rho * m;
                        }
                        sG[i * S + lane] = t;
                    }
                __syncthreads();
                invert_sG();
            }
            // ---- the QP of node u in wave 0 (k_lg_qp_setup + k_lg_build_winv + the ADMM of k_lg_admm_wave)
            if (wave == 0) {
                const double kuu = sKT[u * S + u], guu = GENERAL ? 1.0 : sG[u * S + u];
                const bool active = col && lane != u && (!GENERAL || s_a[col ? lane : 0] != 0.0);
                const double pi = active ? sKT[u * S + lane] / kuu : 0.0;
                const double gi = (col && !GENERAL) ? sG[lane * S + u] : 0.0;
                double w[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    double t = 0.0;
                    if (col && j < n) {
                        if (GENERAL) {
                            t = sG[lane * S + j];
                        } else {
                            t = (lane == j) ? 1.0 : 0.0;
                            if (lane != u && j != u) t -= rho * (sG[lane * S + j] - (gi * sG[j * S + u]) / guu);
                        }
                    }
                    w[j] = t;
                }
                double xo = 0, zo = 0, uo = 0;
                double v = active ? (0.0 - 0.0) + rho * pi : 0.0;
                for (int it = 0; it < max_qp_iter; ++it) {
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four independent fused chains (see k_lg_admm_wave)
#pragma unroll
                    for (int j = 0; j < NP; j += 4) {
                        a0 = __builtin_fma(w[j], lg_bcast(v, j), a0);
                        a1 = __builtin_fma(w[j + 1], lg_bcast(v, j + 1), a1);
                        a2 = __builtin_fma(w[j + 2], lg_bcast(v, j + 2), a2);
                        a3 = __builtin_fma(w[j + 3], lg_bcast(v, j + 3), a3);
                    }
                    const double acc = (a0 + a1) + (a2 + a3);
                    const double xn = active ? acc : 0.0;
                    double zn = xn + uo;
                    zn = zn < 0 ? 0.0 : zn;
                    const double un = uo + (xn - zn);
                    double dx2 = 0, x2 = 0, xm = 0, dz2 = 0, z2 = 0, zm = 0;
                    if (active) {
                        dx2 = (xn - xo) * (xn - xo);
                        x2 = xo * xo;
                        xm = fabs(xo);
                        dz2 = (zn - zo) * (zn - zo);
                        z2 = zo * zo;
                        zm = fabs(zo);
                    }
                    xo = xn;
                    zo = active ? zn : 0.0;
                    uo = active ? un : 0.0;
                    v = active ? (zn - un) + rho * pi : 0.0;
                    if (it > 0) {
                        double sq[4] = {dx2, x2, dz2, z2};
                        lg_wsum4(sq, lane);
                        const bool xs = lg_prev_is_zero(sq[1], 64.0, [&] { return wmax(xm); });
                        const bool zs = lg_prev_is_zero(sq[3], 64.0, [&] { return wmax(zm); });
                        if (lg_rule_met2(sq[0], sq[1], xs, sq[2], sq[3], zs, D.qp_tol)) break;
                    }
                }
                if (col) {
                    s_beta[lane] = active ? -zo : 0.0;  // k_lg_beta
                    s_h[lane] = sC[lane * S + u];       // k_lg_snap_col: the old column of C
                }
            }
            __syncthreads();
            // ---- t = C beta: one wave per row, the order of k_matvec
            for (int i = wave; i < n; i += 4) {
                double acc = col ? sC[i * S + lane] * s_beta[lane] : 0.0;
                acc = wsum(acc);
                if (lane == 0) s_t[i] = acc;
            }
            __syncthreads();
            // ---- k_lg_node_update in wave 0: M beta, the new row / column of Theta, the change vector a
            if (wave == 0) {
                const double cuu = sC[u * S + u], kuu = sKT[u * S + u];
                const double ciu = col ? s_h[lane] : 0.0;
                const double bi = col ? s_beta[lane] : 0.0;
                const double cb = wsum((col && lane != u) ? ciu * bi : 0.0);
                double ti = 0.0;
                if (col && lane != u) ti = s_t[lane] - ciu * cb / cuu;
                const double quad = wsum(bi * ti);
                const double th_uu_old = sTh[u * S + u];
                const double th_uu_new = 1.0 / kuu + quad;
                if (col) {
                    s_t[lane] = ti;
                    if (lane == u) {
                        s_a[lane] = 0.5 * (th_uu_new - th_uu_old);
                    } else {
                        s_a[lane] = bi - sTh[lane * S + u];
                        sTh[lane * S + u] = bi;
                        sTh[u * S + lane] = bi;
                    }
                }
                if (lane == 0) {
                    sTh[u * S + u] = th_uu_new;
                    s_sc[0] = kuu;
                    s_sc[1] = cuu;
                }
            }
            __syncthreads();
            // ---- k_lg_update_c (element-wise) and h = G a (one wave per row)
            {
                const double kuu = s_sc[0], cuu = s_sc[1];
                for (int i = wave; i < n; i += 4)
                    if (col) {
                        const int j = lane;
                        const double gi = s_t[i] * kuu, gj = s_t[j] * kuu;
                        double o;
                        if (i == u && j == u)
                            o = kuu;
                        else if (i == u)
                            o = -gj;
                        else if (j == u)
                            o = -gi;
                        else {
                            const double m = sC[i * S + j] - (s_h[i] * s_h[j]) / cuu;
                            o = m + (gi * gj) / kuu;
                        }
                        sC[i * S + j] = o;
                    }
            }
            __syncthreads();  // every reader of the snapshot h is done: h becomes G a (one wave per row)
            if (GENERAL) continue;  // no G to maintain: W is rebuilt from C for the next node
            for (int i = wave; i < n; i += 4) {
                double acc = col ? sG[i * S + lane] * s_a[lane] : 0.0;
                acc = wsum(acc);
                if (lane == 0) s_h[i] = acc;
            }
            __syncthreads();
            // ---- k_lg_wood_scal in wave 0, then k_lg_update_g (element-wise)
            if (wave == 0) {
                const double gci = col ? sG[lane * S + u] : 0.0;  // g = G e_u
                const double ai = col ? s_a[lane] : 0.0, hi = col ? s_h[lane] : 0.0;
                const double ag = wsum(ai * gci), ah = wsum(ai * hi);
                if (col) s_g[lane] = gci;
                if (lane == 0) {
                    const double s11 = 1.0 + ag, s12 = ah, s21 = sG[u * S + u], s22 = 1.0 + s_h[u];
                    const double det = s11 * s22 - s12 * s21;
                    s_sc[4] = s22 / det;  // inverse of S
                    s_sc[5] = -s12 / det;
                    s_sc[6] = -s21 / det;
                    s_sc[7] = s11 / det;
                }
            }
            __syncthreads();
            {
                const double i11 = s_sc[4], i12 = s_sc[5], i21 = s_sc[6], i22 = s_sc[7];
                for (int i = wave; i < n; i += 4)
                    if (col) {
                        const int j = lane;
                        const double gi = s_g[i], hi = s_h[i], gj = s_g[j], hj = s_h[j];
                        const double corr = gi * (i11 * hj + i12 * gj) + hi * (i21 * hj + i22 * gj);
                        sG[i * S + j] -= corr;
                    }
            }
            __syncthreads();
        }
        cycles = cycle + 1;
        if (cycle > 4) {  // :221  relative_error(l_pre, Theta) (evaluation.py:21-25)
            double d2 = 0, a2 = 0, am = 0;
            for (int i = wave; i < n; i += 4)
                if (col) {
                    const double a = D.ThPrev[(int64_t)i * n + lane], b = sTh[i * S + lane];
                    d2 += (b - a) * (b - a);
                    a2 += a * a;
                    am = fmax(am, fabs(a));
                }
            d2 = wsum(d2);
            a2 = wsum(a2);
            am = wmax(am);
            if (lane == 0) {
                s_red[0][wave] = d2;
                s_red[1][wave] = a2;
                s_red[2][wave] = am;
            }
            __syncthreads();
            d2 = ((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3];
            a2 = ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3];
            am = fmax(fmax(s_red[2][0], s_red[2][1]), fmax(s_red[2][2], s_red[2][3]));
            const double rel = (am <= 1e-8) ? sqrt(d2) : sqrt(d2) / sqrt(a2);
            __syncthreads();
            if (rel < prob_tol) {
                converged = cycle;
                break;
            }
        }
    }
    for (int i = wave; i < n; i += 4)
        if (col) {
            D.Th[(int64_t)i * n + lane] = sTh[i * S + lane];
            D.C[(int64_t)i * n + lane] = sC[i * S + lane];
        }
    if (tid == 0) {
        out[0] = (double)converged;
        out[1] = (double)cycles;
    }
}

struct AdmmRes {
    bool one = false;  // n <= 64: single-workgroup form (k_lg_admm_one)
    bool wave = false;  // n <= 64: single-wave form (k_lg_admm_wave), the default there
    bool quad = false;  // 64 < n <= 160: one workgroup, four lanes per row, rows in registers (k_lg_admm_quad), the default there
    bool on = false;
    int R = 0, nwg = 0;
    size_t lds = 0;
    unsigned *bar = nullptr;
    // "tagged" form (k_lg_admm_tag): v travels as self-validating records, no grid barrier
    bool tagged = false;
    bool wide = false;      // tagged form with 512 threads and up to 16 rows per workgroup (8 of them in registers)
    bool regs = false;      // tagged form with 512 threads and 16 rows per workgroup, all of them in registers
    bool regs24 = false;    // ... and eight more in LDS
    lg_u4 *gran = nullptr;  // [2][n] records of 16 bytes
    unsigned next_tag = 0;  // first tag of the next launch (a launch uses max_iter + 1 consecutive ones)
};

// Which form the QP's ADMM loop takes.  GRAPHLEARN_AMD_LGMRF_ADMM = launch | wave | quad | one | resident | tagged forces
// one (where it applies).  Measured per node-update, one cycle of the config-3 input (every QP runs its
// 100 steps), launch / grid-barrier / tagged:  n = 128: 0.66 / 0.61 / 0.38 ms,  256: 0.67 / 0.72 / 0.39,
// 500: 0.90 / 0.81 / 0.46,  1024: 0.96 / 0.89 / 0.56,  2048: 1.62 / 1.20 / 0.80;  single workgroup
// against tagged at n = 32: 0.26 / 0.20, n = 64: 0.48 / 0.30.  To n = 64 the default is the single-wave form.

int admm_res_plan(gl_ctx *ctx, int64_t n, AdmmRes &A, bool set_lds_limit = true) {
    A.on = A.one = A.wave = A.quad = A.tagged = false;
    const char *force = getenv("GRAPHLEARN_AMD_LGMRF_ADMM");
    auto is = [&](const char *v) { return force && !strcmp(force, v); };
    if (is("launch")) return GL_OK;
    if (n <= ADMM_ONE_MAXN && (is("wave") || !force)) {  // one wave, rows in registers
        A.wave = true;
        return GL_OK;
    }
    if (n > ADMM_ONE_MAXN && n <= ADMM_QUAD_MAXN && (is("quad") || is("wave") || !force)) {  // one workgroup, rows in registers
        A.quad = true;
        return GL_OK;
    }
    if (n <= ADMM_ONE_MAXN && is("one")) {
        A.lds = ((size_t)n * n + 2 * n + 6 * 16) * 8;
        if (set_lds_limit) GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_admm_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A.lds));
        A.one = true;
        return GL_OK;
    }
    const int ncu = ctx->prop.multiProcessorCount;
    int R = (int)((n + ncu - 1) / ncu);
    // a few rows per workgroup even when there are CUs to spare: fewer parties in the hand-off of v
    // (n = 256: 1 / 2 / 4 / 8 rows: 0.455 / 0.399 / 0.385 / 0.397 ms per node-update)
    R = std::max<int>(R, (int)std::min<int64_t>(4, (n + 63) / 64));
    if (const char *e = getenv("GRAPHLEARN_AMD_LGMRF_R")) R = std::max(R, std::min(ADMM_RES_MAXR, atoi(e)));  // tuning aid
    size_t lds = (size_t)R * n * 8;
    if (R > ADMM_RES_MAXR || lds > 150 * 1024) return GL_OK;  // one launch per step
    A.tagged = !is("resident");  // "resident" keeps the grid-barrier form (tests, comparison)
    // wide form: twice the rows per workgroup, half the parties in the hand-off (default above n = 1024, where the poll of
    // a step is what a step costs; GRAPHLEARN_AMD_LGMRF_WIDE = 0 | 1 forces)
    // (ms per node update, narrow / wide: n = 256: 0.31 / 0.29, 384: 0.34 / 0.30, 512: 0.33 / 0.22-0.26, 1024: 0.42 / 0.37,
    //  1536: 0.55 / 0.44, 2048: 0.64 / 0.49; 24 rows, 16 of them in registers, measured too: 0.29 / 0.42 / 0.56 at 512 /
    //  1024 / 2048 -- the registers are gone)
    A.wide = A.tagged && n >= 256;
    if (const char *e = getenv("GRAPHLEARN_AMD_LGMRF_WIDE")) A.wide = A.tagged && n > 2 * ADMM_RES_MAXR && atoi(e) != 0;
    if (A.wide) {
        R = 2 * ADMM_RES_MAXR;
        lds = (size_t)ADMM_RES_MAXR * n * 8;
    }
    // "regs" (round 4, the default wherever the wide form applies): its sixteen rows ALL in registers -- the multiply reads no
    // LDS (128 KB per step and workgroup before): 0.467 -> 0.362 ms per node update at n = 2048 (GRAPHLEARN_AMD_LGMRF_REGS = 0 | 1;
    // 2: eight more rows in LDS, 24 per workgroup)
    A.regs = A.wide;
    A.regs24 = false;
    if (const char *e = getenv("GRAPHLEARN_AMD_LGMRF_REGS")) {
        A.regs = A.wide && atoi(e) != 0;
        A.regs24 = A.regs && atoi(e) == 2 && n > 3 * ADMM_RES_MAXR;
    }
    if (A.regs) lds = 0;
    if (A.regs24) {
        R = 3 * ADMM_RES_MAXR;
        lds = (size_t)ADMM_RES_MAXR * n * 8;
    }
    A.R = R;
    A.nwg = (int)((n + R - 1) / R);
    A.lds = lds;
    if (set_lds_limit)
        GL_HIP(ctx, hipFuncSetAttribute(A.regs24   ? (const void *)k_lg_admm_tag<512, 16, 8>
                                        : A.regs   ? (const void *)k_lg_admm_tag<512, 16, 0>
                                        : A.wide   ? (const void *)k_lg_admm_tag<512, 8>
                                        : A.tagged ? (const void *)k_lg_admm_tag<256, 0>
                                                   : (const void *)k_lg_admm_res,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    A.on = true;
    return GL_OK;
}

int run_admm_wave(gl_ctx *ctx, const LgDev &D, int max_iter, int64_t u_fast = -1) {
    if (D.n <= 16)
        k_lg_admm_wave<16><<<1, 64, 0, ctx->stream>>>(D, max_iter, u_fast);
    else if (D.n <= 32)
        k_lg_admm_wave<32><<<1, 64, 0, ctx->stream>>>(D, max_iter, u_fast);
    else
        k_lg_admm_wave<64><<<1, 64, 0, ctx->stream>>>(D, max_iter, u_fast);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

int run_admm_quad(gl_ctx *ctx, const LgDev &D, int max_iter, int64_t u_fast = -1) {
    if (D.n <= 128)
        k_lg_admm_quad<32><<<1, 512, 0, ctx->stream>>>(D, max_iter, u_fast);
    else
        k_lg_admm_quad<40><<<1, 640, 0, ctx->stream>>>(D, max_iter, u_fast);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

int run_admm_one(gl_ctx *ctx, const LgDev &D, const AdmmRes &A, int max_iter) {
    k_lg_admm_one<<<1, 1024, A.lds, ctx->stream>>>(D, max_iter);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

int run_admm_res(gl_ctx *ctx, const LgDev &D, AdmmRes &A, int max_iter, int64_t u_fast = -1) {
    const bool test_not_resident = getenv("GRAPHLEARN_AMD_LGMRF_TEST_NOT_RESIDENT") != nullptr;  // tests: the rendezvous can never complete
    if (!A.tagged) GL_HIP(ctx, hipMemsetAsync(A.bar, 0, GL_BAR_WORDS * 4, ctx->stream));
    const GlGridBar gb{A.bar, A.nwg + (test_not_resident ? 1 : 0)};
    if (A.tagged) {
        // tags of this launch: next_tag .. next_tag + max_iter (never 0, never one a record may still carry)
        if (max_iter >= (1 << 28)) return gl_set_error(ctx, GL_E_INVALID, "ADMM: max_iter too large");
        if (A.next_tag == 0u || A.next_tag > 0xE0000000u) {  // first use, or the tags would wrap into old ones
            GL_HIP(ctx, hipMemsetAsync(A.gran, 0, (size_t)2 * D.n * 16, ctx->stream));
            A.next_tag = 1u;
        }
        const unsigned tag0 = A.next_tag;
        A.next_tag += (unsigned)max_iter + 2u;
        // error word: bar[32], zero since the allocation (a failure ends the use of this form)
        static const int poll_mode = getenv("GRAPHLEARN_AMD_LGMRF_POLL") ? atoi(getenv("GRAPHLEARN_AMD_LGMRF_POLL")) : 0;
        const int flags = (test_not_resident ? 1 : 0) | (poll_mode << 1);
        if (A.regs24)
            k_lg_admm_tag<512, 16, 8><<<dim3((unsigned)A.nwg), 512, A.lds, ctx->stream>>>(D, A.bar + 32, A.gran, tag0, A.R, max_iter, u_fast, flags);
        else if (A.regs)
            k_lg_admm_tag<512, 16, 0><<<dim3((unsigned)A.nwg), 512, A.lds, ctx->stream>>>(D, A.bar + 32, A.gran, tag0, A.R, max_iter, u_fast, flags);
        else if (A.wide)
            k_lg_admm_tag<512, 8><<<dim3((unsigned)A.nwg), 512, A.lds, ctx->stream>>>(D, A.bar + 32, A.gran, tag0, A.R, max_iter, u_fast, flags);
        else
            k_lg_admm_tag<256, 0><<<dim3((unsigned)A.nwg), 256, A.lds, ctx->stream>>>(D, A.bar + 32, A.gran, tag0, A.R, max_iter, u_fast, flags);
        GL_HIP(ctx, hipGetLastError());
        return GL_OK;
    }
    k_lg_admm_res<<<dim3((unsigned)A.nwg), 256, A.lds, ctx->stream>>>(D, gb, A.R, max_iter);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

int run_admm(gl_ctx *ctx, const LgDev &D, int max_iter) {
    const unsigned nblk = (unsigned)((D.n + ADMM_ROWS - 1) / ADMM_ROWS);
    double *flag = (double *)ctx->pinned + 32;
    int it = 0;
    int chunk = 24;
    while (it < max_iter) {
        const int end = std::min(max_iter, it + chunk);
        for (; it < end; ++it) k_lg_admm<<<nblk, 256, 0, ctx->stream>>>(D, it);
        GL_HIP(ctx, hipGetLastError());
        if (it >= max_iter) break;
        GL_HIP(ctx, hipMemcpyAsync(flag, D.scal, 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (*flag != 0.0) break;
        chunk = 16;
    }
    return GL_OK;
}

// after the loop: the result z is in D.z; the next right-hand side buffers are scratch again.

// beta = -z ; prepares the M beta product: (M beta)_i = (C beta)_i - c_i (c.beta)/c_uu
__global__ __launch_bounds__(VT) void k_lg_beta(LgDev D, int64_t u) {
    for (int64_t i = threadIdx.x; i < D.n; i += VT) D.beta[i] = (D.act[i] != 0.0) ? -D.z[i] : 0.0;
}

// after t = C beta:  t <- M beta (entry u = 0), Theta row/col u, the C row/col u, and the
// Theta-change vector `a` for the Woodbury correction of G   (lgmrf.py:207-215)
__global__ __launch_bounds__(VT) void k_lg_node_update(LgDev D, int64_t u) {
    __shared__ double sh[VT / 64];
    const int64_t n = D.n;
    const double cuu = D.C[u * n + u], kuu = D.K[u * n + u];
    double part = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT)
        if (i != u) part += D.h[i] * D.beta[i];  // h = column u of C (snapshot taken before: contiguous, not a strided walk)
    const double cb = block_sum(part, sh);
    double quad = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT) {
        double ti = 0.0;
        if (i != u) ti = D.t[i] - D.h[i] * cb / cuu;
        D.t[i] = ti;  // (M beta)_i
        quad += D.beta[i] * ti;
    }
    quad = block_sum(quad, sh);
    const double th_uu_old = D.Th[u * n + u];
    const double th_uu_new = 1.0 / kuu + quad;
    for (int64_t i = threadIdx.x; i < n; i += VT) {
        if (i == u) {
            D.a[i] = 0.5 * (th_uu_new - th_uu_old);
            continue;
        }
        D.a[i] = D.beta[i] - D.Th[i * n + u];
        D.Th[i * n + u] = D.beta[i];
        D.Th[u * n + i] = D.beta[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        D.Th[u * n + u] = th_uu_new;
        D.scal[2] = kuu;
        D.scal[3] = cuu;
    }
}

// C[m,m] = (C[m,m] - c c'/c_uu) + g g'/k_uu with g = (M beta) k_uu; C[u,m] = C[m,u] = -g; C[u,u] = k_uu.
// Needs the OLD column c for every entry, so the old column is snapshot into `h` first.
__global__ __launch_bounds__(VT) void k_lg_snap_col(LgDev D, int64_t u) {
    for (int64_t i = threadIdx.x; i < D.n; i += VT) D.h[i] = D.C[i * D.n + u];
}
__global__ void k_lg_update_c(LgDev D, int64_t u) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = D.n;
    if (j >= n) return;
    const double kuu = D.scal[2], cuu = D.scal[3];
    const double gi = D.t[i] * kuu, gj = D.t[j] * kuu;
    double out;
    if (i == u && j == u)
        out = kuu;
    else if (i == u)
        out = -gj;
    else if (j == u)
        out = -gi;
    else {
        const double m = D.C[i * n + j] - (D.h[i] * D.h[j]) / cuu;  // lapl_u_inv (lgmrf.py:186-188)
        out = m + (gi * gj) / kuu;                                  // Sherman-Morrison (:217-219)
    }
    D.C[i * n + j] = out;
}

// Woodbury rank-2 correction of G = (Theta + rho I)^-1 after Theta changed by e_u a' + a e_u':
//   G <- G - [g h] S^-1 [h g]',  S = [[1 + a.g, a.h], [G_uu, 1 + h_u]],  g = G e_u, h = G a.
__global__ __launch_bounds__(VT) void k_lg_wood_scal(LgDev D, int64_t u) {
    __shared__ double sh[VT / 64];
    const int64_t n = D.n;
    double ag = 0, ah = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT) {
        const double gi = D.G[i * n + u];
        D.gcol[i] = gi;
        ag += D.a[i] * gi;
        ah += D.a[i] * D.h[i];
    }
    ag = block_sum(ag, sh);
    ah = block_sum(ah, sh);
    if (threadIdx.x == 0) {
        const double s11 = 1.0 + ag, s12 = ah, s21 = D.G[u * n + u], s22 = 1.0 + D.h[u];
        const double det = s11 * s22 - s12 * s21;
        D.scal[4] = s22 / det;   // inverse of S
        D.scal[5] = -s12 / det;
        D.scal[6] = -s21 / det;
        D.scal[7] = s11 / det;
    }
}
__global__ void k_lg_update_g(LgDev D) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = D.n;
    if (j >= n) return;
    const double i11 = D.scal[4], i12 = D.scal[5], i21 = D.scal[6], i22 = D.scal[7];
    const double gi = D.gcol[i], hi = D.h[i], gj = D.gcol[j], hj = D.h[j];
    // [g_i h_i] S^-1 [h_j ; g_j]
    const double corr = gi * (i11 * hj + i12 * gj) + hi * (i21 * hj + i22 * gj);
    D.G[i * n + j] -= corr;
}

// ---- the node update around the QP in two launches per matrix instead of three (round 4) ----
// K transposed once per fit (the fused QP set-up reads column u of K as a row)
__global__ void k_lg_transpose(const double *__restrict__ A, double *__restrict__ AT, int64_t n) {
    __shared__ double tile[32][33];
    const int64_t bx = (int64_t)blockIdx.x * 32, by = (int64_t)blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = by + r, j = bx + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && j < n) ? A[i * n + j] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = bx + r, j = by + threadIdx.x;
        if (i < n && j < n) AT[i * n + j] = tile[threadIdx.x][r];
    }
}
// column u of G (the first node of a cycle; afterwards k_lg_update_g2 writes the next node's column as it goes)
__global__ __launch_bounds__(VT) void k_lg_gcol(LgDev D, int64_t u) {
    for (int64_t i = threadIdx.x; i < D.n; i += VT) D.gcol[i] = D.G[i * D.n + u];
}
constexpr int LG_RB = 8;  // rows per workgroup of the fused updates (n = 2048: 256 workgroups of 16 waves)
// k_lg_node_update + k_lg_update_c in ONE launch.  Every workgroup forms the two sums of k_lg_node_update itself (same thread
// count, same order: the same bits) -- c'beta over the snapshot h of column u, and beta'(M beta) --, derives
// (M beta)_j = t_j - h_j (c'beta) / c_uu on the fly for its rows and columns and updates LG_RB rows of C; workgroup 0 also
// writes Theta's row / column u, Theta_uu and the change vector `a`.  c_uu comes from the snapshot (h_u): C_uu itself is
// rewritten by the workgroup of row u during this launch.
__global__ __launch_bounds__(VT) void k_lg_update_c2(LgDev D, int64_t u) {
    __shared__ double sh[VT / 64];
    const int64_t n = D.n;
    const double cuu = D.h[u], kuu = D.K[u * n + u];
    double part = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT)
        if (i != u) part += D.h[i] * D.beta[i];
    const double cb = block_sum(part, sh);
    auto mbeta = [&](int64_t i) { return i != u ? D.t[i] - D.h[i] * cb / cuu : 0.0; };  // (M beta)_i  (t = C beta)
    double quad = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT) quad += D.beta[i] * mbeta(i);
    quad = block_sum(quad, sh);
    if (blockIdx.x == 0) {
        const double th_uu_old = D.Th[u * n + u];
        const double th_uu_new = 1.0 / kuu + quad;
        for (int64_t i = threadIdx.x; i < n; i += VT) {
            if (i == u) {
                D.a[i] = 0.5 * (th_uu_new - th_uu_old);
                continue;
            }
            D.a[i] = D.beta[i] - D.Th[u * n + i];  // (Theta is symmetric bit for bit: row and column u are written together)
            D.Th[i * n + u] = D.beta[i];
            D.Th[u * n + i] = D.beta[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) D.Th[u * n + u] = th_uu_new;
    }
    const int64_t i0 = (int64_t)blockIdx.x * LG_RB;
    __shared__ double s_gi[LG_RB], s_hi[LG_RB];  // (one division per row and workgroup, not per entry)
    if (threadIdx.x < LG_RB && i0 + threadIdx.x < n) {
        s_gi[threadIdx.x] = mbeta(i0 + threadIdx.x) * kuu;
        s_hi[threadIdx.x] = D.h[i0 + threadIdx.x];
    }
    __syncthreads();
    auto entry = [&](int64_t i, int64_t j, double cij, double gi, double hi, double gj, double hj) {
        if (i == u && j == u) return kuu;
        if (i == u) return -gj;
        if (j == u) return -gi;
        const double m = cij - (hi * hj) / cuu;  // lapl_u_inv (lgmrf.py:186-188)
        return m + (gi * gj) / kuu;              // Sherman-Morrison (:217-219)
    };
    if ((n & 1) == 0) {  // rows start 16-byte aligned: two columns per access
        for (int64_t j = 2 * (int64_t)threadIdx.x; j < n; j += 2 * VT) {
            const double gj0 = mbeta(j) * kuu, hj0 = D.h[j], gj1 = mbeta(j + 1) * kuu, hj1 = D.h[j + 1];
#pragma unroll
            for (int r = 0; r < LG_RB; ++r) {
                const int64_t i = i0 + r;
                if (i >= n) break;
                double2 *p = reinterpret_cast<double2 *>(D.C + i * n + j);
                const double2 c = *p;
                *p = double2{entry(i, j, c.x, s_gi[r], s_hi[r], gj0, hj0), entry(i, j + 1, c.y, s_gi[r], s_hi[r], gj1, hj1)};
            }
        }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += VT) {
            const double gj = mbeta(j) * kuu, hj = D.h[j];
#pragma unroll
            for (int r = 0; r < LG_RB; ++r) {
                const int64_t i = i0 + r;
                if (i >= n) break;
                D.C[i * n + j] = entry(i, j, D.C[i * n + j], s_gi[r], s_hi[r], gj, hj);
            }
        }
    }
}
// k_lg_wood_scal + k_lg_update_g in ONE launch, after h = G a: every workgroup forms the two sums of k_lg_wood_scal itself
// (same thread count, same order) and the inverse of the 2 x 2 matrix S, corrects LG_RB rows of G and leaves column u_next of
// the corrected G in gcol_next for the next node's QP (gcol, column u before the correction, is read by everybody throughout:
// two buffers).  G_uu comes from the snapshot (gcol_u).
__global__ __launch_bounds__(VT) void k_lg_update_g2(LgDev D, int64_t u, int64_t u_next) {
    __shared__ double sh[VT / 64];
    const int64_t n = D.n;
    double ag = 0, ah = 0;
    for (int64_t i = threadIdx.x; i < n; i += VT) {
        ag += D.a[i] * D.gcol[i];
        ah += D.a[i] * D.h[i];
    }
    ag = block_sum(ag, sh);
    ah = block_sum(ah, sh);
    const double s11 = 1.0 + ag, s12 = ah, s21 = D.gcol[u], s22 = 1.0 + D.h[u];
    const double det = s11 * s22 - s12 * s21;
    const double i11 = s22 / det, i12 = -s12 / det, i21 = -s21 / det, i22 = s11 / det;  // inverse of S
    const int64_t i0 = (int64_t)blockIdx.x * LG_RB;
    auto corr = [&](double gi, double hi, double gj, double hj) { return gi * (i11 * hj + i12 * gj) + hi * (i21 * hj + i22 * gj); };  // [g_i h_i] S^-1 [h_j ; g_j]
    if ((n & 1) == 0) {  // rows start 16-byte aligned: two columns per access
        for (int64_t j = 2 * (int64_t)threadIdx.x; j < n; j += 2 * VT) {
            const double gj0 = D.gcol[j], hj0 = D.h[j], gj1 = D.gcol[j + 1], hj1 = D.h[j + 1];
#pragma unroll
            for (int r = 0; r < LG_RB; ++r) {
                const int64_t i = i0 + r;
                if (i >= n) break;
                const double gi = D.gcol[i], hi = D.h[i];
                double2 *p = reinterpret_cast<double2 *>(D.G + i * n + j);
                const double2 g = *p;
                const double2 gn = double2{g.x - corr(gi, hi, gj0, hj0), g.y - corr(gi, hi, gj1, hj1)};
                *p = gn;
                if (j == u_next) D.gcol_next[i] = gn.x;
                if (j + 1 == u_next) D.gcol_next[i] = gn.y;
            }
        }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += VT) {
            const double gj = D.gcol[j], hj = D.h[j];
#pragma unroll
            for (int r = 0; r < LG_RB; ++r) {
                const int64_t i = i0 + r;
                if (i >= n) break;
                const double gn = D.G[i * n + j] - corr(D.gcol[i], D.h[i], gj, hj);
                D.G[i * n + j] = gn;
                if (j == u_next) D.gcol_next[i] = gn;
            }
        }
    }
}

// G0 = Theta + rho I (then inverted in place by Gauss-Jordan)
__global__ void k_lg_shift(const double *__restrict__ Th, double *__restrict__ G, int64_t n, double rho) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) G[i * n + j] = Th[i * n + j] + (i == j ? rho : 0.0);
}

// partial sums of ||a - b||^2, ||a||^2 and max|a| over a flat array (deterministic two-stage reduce)
__global__ __launch_bounds__(256) void k_relerr_part(const double *__restrict__ a, const double *__restrict__ b,
                                                     int64_t count, double *__restrict__ part) {
    __shared__ double sh[4];
    double d2 = 0, a2 = 0, am = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const double da = b[i] - a[i];
        d2 += da * da;
        a2 += a[i] * a[i];
        am = fmax(am, fabs(a[i]));
    }
    d2 = block_sum(d2, sh);
    a2 = block_sum(a2, sh);
    am = block_max(am, sh);
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x + 0] = d2;
        part[3 * blockIdx.x + 1] = a2;
        part[3 * blockIdx.x + 2] = am;
    }
}

// numpy's pairwise summation (np.add.reduce on a contiguous float64 array), exactly
double np_pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double s = 0;
        for (int64_t i = 0; i < n; ++i) s += a[i];
        return s;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) s += a[i];
        return s;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// ---- general path, compact form.  The QP of node u only involves the entries its column of adj_mask allows
// (lgmrf.py:191-202: M[nz][:, nz], k_u[nz]); the reference gathers exactly that submatrix.  So does this path: the host
// knows every node's active list (the mask never changes), the device gathers W = I + rho M[nz, nz] (na x na, M = C -
// c c'/c_uu formed on the fly), inverts THAT (one launch up to na = 128) and runs the ADMM on vectors of length na, with
// whatever form of the loop fits na.  A sparse mask makes every QP tiny whatever n is; a full mask costs what the
// n x n formulation cost (which inverted an identity block for the inactive entries).  Active entries keep their
// ascending order, so eliminations and sums run in the reference's order.
__global__ void k_lg_compact_w(LgDev D, LgDev Dc, int64_t u, const int *__restrict__ idx) {
    const int64_t a = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = D.n, na = Dc.n;
    if (b >= na) return;
    const int64_t i = idx[a], j = idx[b];
    const double m = D.C[i * n + j] - (D.C[i * n + u] * D.C[j * n + u]) / D.C[u * n + u];
    Dc.W[a * na + b] = ((a == b) ? 1.0 : 0.0) + D.rho * m;
}
// p = k_u[nz] / k_uu, ADMM start (x = z = u = 0, v = rho p)   (lgmrf.py:197-201, :55); na may be 0
__global__ __launch_bounds__(VT) void k_lg_compact_vecs(LgDev D, LgDev Dc, int64_t u, const int *__restrict__ idx) {
    const double kuu = D.K[u * D.n + u];
    for (int64_t a = threadIdx.x; a < Dc.n; a += VT) {
        const double pa = D.K[(int64_t)idx[a] * D.n + u] / kuu;
        Dc.act[a] = 1.0;
        Dc.p[a] = pa;
        Dc.x[a] = 0.0;
        Dc.z[a] = 0.0;
        Dc.uu[a] = 0.0;
        Dc.v[a] = (0.0 - 0.0) + D.rho * pa;
    }
    if (threadIdx.x == 0) {
        D.scal[0] = 0.0;
        D.scal[1] = 0.0;
    }
}
// beta = -z on the active entries, 0 elsewhere (lgmrf.py:197-209)
__global__ __launch_bounds__(VT) void k_lg_scatter_beta(LgDev D, LgDev Dc, const int *__restrict__ idx) {
    for (int64_t i = threadIdx.x; i < D.n; i += VT) D.beta[i] = 0.0;
    __syncthreads();
    for (int64_t a = threadIdx.x; a < Dc.n; a += VT) D.beta[idx[a]] = -Dc.z[a];
}

struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
};

// The same Gauss-Jordan for n <= 128 in ONE launch of ONE workgroup, the matrix in REGISTERS: thread (wave w,
// lane l) owns the entries (w + 16 r, l + 64 c), r < 8, c < 2; per pivot the scaled pivot row and the pivot
// column go through LDS (double-buffered: one workgroup barrier per pivot) and every thread updates its 16
// entries from 10 LDS reads.  n = 64: 128 launches = 0.6 ms -> ~30 us.  Element-wise the arithmetic of
// k_gj_pivot / k_gj_eliminate, so the result is bit-identical.
constexpr int GJ_LDS_MAXN = 128;
__global__ __launch_bounds__(1024) void k_gj_lds(double *__restrict__ A, int n) {
    __shared__ double s_prow[2][GJ_LDS_MAXN], s_pcol[2][GJ_LDS_MAXN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = wave + 16 * r, j = lane + 64 * c;
            a[r][c] = (i < n && j < n) ? A[(int64_t)i * n + j] : 0.0;
        }
    for (int k = 0; k < n; ++k) {
        double *prow = s_prow[k & 1], *pcol = s_pcol[k & 1];
        const int kr = k >> 4, kw = k & 15, kc = k >> 6, kl = k & 63;  // owner of row k: wave kw, slot kr; of column k: lane kl, slot kc
        if (wave == kw) {  // scaled pivot row
            double rowv[2] = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r == kr) {
                    rowv[0] = a[r][0];
                    rowv[1] = a[r][1];
                }
            const double mine = kc ? rowv[1] : rowv[0];
            const long long pb = __double_as_longlong(mine);
            const double piv = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(pb >> 32), kl) << 32) |
                                                    (unsigned)__builtin_amdgcn_readlane((int)pb, kl));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = lane + 64 * c;
                if (j < n) prow[j] = (j == k) ? 1.0 / piv : rowv[c] / piv;
            }
        }
        if (lane == kl) {  // pivot column before it is overwritten
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = wave + 16 * r;
                if (i < n) pcol[i] = kc ? a[r][1] : a[r][0];
            }
        }
        __syncthreads();
        double pr[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) pr[c] = (lane + 64 * c < n) ? prow[lane + 64 * c] : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = wave + 16 * r;
            if (i >= n) continue;
            const double pc = pcol[i];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = lane + 64 * c;
                if (i == k) {
                    a[r][c] = pr[c];
                } else {
                    const double cur = (j == k) ? 0.0 : a[r][c];
                    a[r][c] = cur - pc * pr[c];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = wave + 16 * r, j = lane + 64 * c;
            if (i < n && j < n) A[(int64_t)i * n + j] = a[r][c];
        }
}

// ---- blocked Gauss-Jordan (n > 128): the same in-place inversion in steps of BG = 64 pivots, two launches per step
// instead of two per pivot (n = 2048: 65 launches instead of 4096).  Step k, K = [64 k, 64 k + 64), P_k = inv(A_KK):
//   k_bgj_panel  R_j = P_k A_Kj -> A_Kj,  C_i = A_iK (old),  A_iK = -C_i P_k              (2 nb workgroups)
//   k_bgj_update A_ij += (-C_i) R_j for i, j != k;  A_KK = P_k;  the workgroup of tile (k + 1, k + 1) goes on to invert
//                its result in registers (unblocked elimination): P_{k+1}, ready when the other tiles are  (nb^2 workgroups)
// (P_0 by k_bgj_pivot0.)  The 64 x 64 x 64 products run on v_mfma_f64_16x16x4_f64: at the same peak as the vector unit on
// this part it takes one operand pair per 16 multiply-adds per lane where a 4 x 4 register block reads eight -- the vector
// form of these kernels was bound by LDS reads (39 us per update; profiles/r03_lgmrf_inverse.md).  No pivoting (the matrices
// are symmetric positive definite); every sum in a fixed order: bit-stable run to run.  The rounding differs from the
// unblocked elimination's (other association, the pivot's reciprocal instead of quotients), as that one's does from LAPACK's.
constexpr int BG = 64, BGS = BG + 2;  // block size; LDS row stride in doubles (operand reads at most 2-way bank conflicts)
typedef double bgj_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double bgj_ld(const double *A, int64_t n, int64_t i, int64_t j) {  // rows / columns >= n: identity
    return (i < n && j < n) ? A[i * n + j] : (i == j ? 1.0 : 0.0);
}
// acc (16 x 16: lane l, element e -> row (l >> 4) + 4 e, column l & 15) += a (16 x 4: lane l -> [l & 15][l >> 4]) b (4 x 16:
// lane l -> [l >> 4][l & 15])
__device__ __forceinline__ bgj_d4 bgj_mfma(double a, double b, bgj_d4 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#else
    return acc;
#endif
}
// 1 / x as the compiler's division forms it before its final correction: v_rcp_f64 and two Newton steps (exact for x = 1,
// the pivots of the identity padding); <= 1 ulp
__device__ __forceinline__ double bgj_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x), e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// A 64 x 64 block inverted in the REGISTERS of one workgroup of NW waves: lane = row, wave w holds columns CW w .. CW w + CW - 1
// of its row (CW = 64 / NW).  Per pivot k the pivot row reaches the lanes through v_readlane (every wave broadcasts its own
// columns of row k from lane k), the pivot column through LDS (64 doubles, double-buffered: one workgroup barrier per pivot).
// The kernel is bound by instruction issue (64 dependent pivots), so the special cases of k_gj_pivot / k_gj_eliminate are
// folded into ONE multiply and ONE multiply-add per entry:  a_ij <- t_i a_kj + m_i a_ij  with  t_i = -a_ik / piv, m_i = 1
// (rows i != k),  t_k = 1 / piv, m_k = 0  (the pivot row: a_kj / piv),  after column k was replaced by the unit vector e_k
// (its entries become -a_ik / piv and 1 / piv).
template <int NW>
__device__ __forceinline__ void bgj_invert_regs(double (&a)[BG / NW], double (*s_col)[BG], int lane, int wave) {
    constexpr int CW = BG / NW;
    for (int kw = 0; kw < NW; ++kw) {
#pragma unroll
        for (int kc = 0; kc < CW; ++kc) {  // (unrolled: the pivot column's register is a compile-time choice)
            const int k = kw * CW + kc;
            if (wave == kw) {
                s_col[kc & 1][lane] = a[kc];
                a[kc] = (lane == k) ? 1.0 : 0.0;
            }
            __syncthreads();
            const double rpiv = bgj_rcp(s_col[kc & 1][k]), pc = s_col[kc & 1][lane];
            const bool prow = lane == k;
            const double m = prow ? 0.0 : 1.0, t = (prow ? 1.0 : -pc) * rpiv;
#pragma unroll
            for (int c = 0; c < CW; ++c) a[c] = __builtin_fma(t, lg_bcast(a[c], k), a[c] * m);
        }
    }
}
__global__ __launch_bounds__(256) void k_bgj_pivot0(const double *__restrict__ A, int64_t n, double *__restrict__ P) {
    __shared__ double s_col[2][BG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = bgj_ld(A, n, lane, 16 * wave + c);
    bgj_invert_regs<4>(a, s_col, lane, wave);
#pragma unroll
    for (int c = 0; c < 16; ++c) P[lane * BG + 16 * wave + c] = a[c];
}
// blockIdx.y = 0: R_b = P A_Kb -> Rbuf and A_Kb;  blockIdx.y = 1: C_b = A_bK (old) -> Cbuf, A_bK = -C_b P.  16 waves, one
// 16 x 16 tile of the product each.
__global__ __launch_bounds__(1024) void k_bgj_panel(double *__restrict__ A, int64_t n, int64_t kb, const double *__restrict__ P,
                                                    double *__restrict__ Rbuf, double *__restrict__ Cbuf) {
    __shared__ double sp[BG][BGS], sx[BG][BGS];  // P; the other operand (A_Kb or A_bK), both row-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool row_part = blockIdx.y == 0;
    const int64_t b = blockIdx.x, k0 = kb * BG, b0 = b * BG, ld = (int64_t)gridDim.x * BG;
    if (b == kb) return;  // (A_KK = P: the update kernel)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 1024 * q, i = e / BG, j = e % BG;
        const double x = row_part ? bgj_ld(A, n, k0 + i, b0 + j) : bgj_ld(A, n, b0 + i, k0 + j);
        sp[i][j] = P[e];
        sx[i][j] = x;
        if (!row_part) Cbuf[b0 * BG + e] = x;                          // C_b: rows b0 .. b0 + 63, 64 columns, row-major
    }
    __syncthreads();
    const int r = lane & 15, q4 = lane >> 4, i0 = 16 * (wave >> 2), j0 = 16 * (wave & 3);
    const double(*sa)[BGS] = row_part ? sp : sx, (*sb)[BGS] = row_part ? sx : sp;
    bgj_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < BG / 4; ++ks) acc = bgj_mfma(sa[i0 + r][4 * ks + q4], sb[4 * ks + q4][j0 + r], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + q4 + 4 * e, j = j0 + r;
        if (row_part) {
            Rbuf[i * ld + b0 + j] = acc[e];  // R: 64 x (nb 64), row-major
            if (k0 + i < n && b0 + j < n) A[(k0 + i) * n + b0 + j] = acc[e];
        } else if (b0 + i < n && k0 + j < n) {
            A[(b0 + i) * n + k0 + j] = -acc[e];
        }
    }
}
// one workgroup per 64 x 64 tile, one wave per 32 x 32 quadrant: the tile's old values are the accumulators' start.  The 64
// pivots of the step are taken in two halves of 32 through 34 KB of LDS (the second half is fetched into registers while the
// first is multiplied), so that four workgroups share a CU and all nb^2 are resident at once: with the whole 64 in LDS (two per
// CU, two rounds) the kernel waited on its loads for 27 us where the products need 6.4.  Block (0, 0) -- the first to be
// scheduled -- is tile (k + 1, k + 1), which then inverts its result for the next step.
constexpr int BGH = BG / 2;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_bgj_update(double *__restrict__ A, int64_t n, int64_t kb, const double *__restrict__ P,
                                                    double *__restrict__ Pnext, const double *__restrict__ Rbuf,
                                                    const double *__restrict__ Cbuf, int nbk) {
    constexpr int SCS = BGH + 2;                               // row stride of the C half
    __shared__ double smem[BG * SCS + BGH * BGS], s_col[2][BG];
    static_assert(BG * SCS + BGH * BGS >= BG * BGS, "the next pivot block is staged in the same LDS");
    double(*sc)[SCS] = reinterpret_cast<double(*)[SCS]>(smem);            // -C_i [i][l - 32 h]
    double(*sr)[BGS] = reinterpret_cast<double(*)[BGS]>(smem + BG * SCS);  // R_j [l - 32 h][j]
    const int64_t ib = (blockIdx.y + kb + 1) % nbk, jb = (blockIdx.x + kb + 1) % nbk, k0 = kb * BG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ib == kb || jb == kb) {
        if (ib == kb && jb == kb)
            for (int e = tid; e < BG * BG; e += 256) {
                const int64_t i = k0 + e / BG, j = k0 + e % BG;
                if (i < n && j < n) A[i * n + j] = P[e];
            }
        return;
    }
    const double *Ci = Cbuf + ib * BG * BG, *Rj = Rbuf + jb * BG;
    const int64_t ldr = (int64_t)nbk * BG;
    double cv[8], rv[8];
    auto fetch = [&](int h) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            cv[q] = Ci[(e / BGH) * BG + BGH * h + e % BGH];
            rv[q] = Rj[(int64_t)(BGH * h + e / BG) * ldr + e % BG];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            sc[e / BGH][e % BGH] = -cv[q];
            sr[e / BG][e % BG] = rv[q];
        }
    };
    fetch(0);
    const int r = lane & 15, q4 = lane >> 4, i0 = 32 * (wave >> 1), j0 = 32 * (wave & 1);
    bgj_d4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t i = ib * BG + i0 + 16 * u + q4 + 4 * e, j = jb * BG + j0 + 16 * v + r;
                acc[u][v][e] = (i < n && j < n) ? A[i * n + j] : 0.0;
            }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();  // (the first half has been read)
        stage();
        __syncthreads();
        if (h == 0) fetch(1);
#pragma unroll 2
        for (int ks = 0; ks < BGH / 4; ++ks) {
            const int l = 4 * ks + q4;
            const double a0 = sc[i0 + r][l], a1 = sc[i0 + 16 + r][l], b0 = sr[l][j0 + r], b1 = sr[l][j0 + 16 + r];
            acc[0][0] = bgj_mfma(a0, b0, acc[0][0]);
            acc[0][1] = bgj_mfma(a0, b1, acc[0][1]);
            acc[1][0] = bgj_mfma(a1, b0, acc[1][0]);
            acc[1][1] = bgj_mfma(a1, b1, acc[1][1]);
        }
    }
    const bool next_pivot = blockIdx.x == 0 && blockIdx.y == 0 && kb + 1 < nbk;  // tile (k + 1, k + 1)
    double(*st)[BGS] = reinterpret_cast<double(*)[BGS]>(smem);
    if (next_pivot) __syncthreads();                                            // (the operands are about to be overwritten)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ti = i0 + 16 * u + q4 + 4 * e, tj = j0 + 16 * v + r;
                const int64_t i = ib * BG + ti, j = jb * BG + tj;
                if (i < n && j < n) A[i * n + j] = acc[u][v][e];
                if (next_pivot) st[ti][tj] = (i < n && j < n) ? acc[u][v][e] : (ti == tj ? 1.0 : 0.0);
            }
    if (!next_pivot) return;
    __builtin_amdgcn_s_setprio(3);  // (64 dependent pivots on the step's critical path, among three other workgroups' MFMAs)
    __syncthreads();
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = st[lane][16 * wave + c];
    bgj_invert_regs<4>(a, s_col, lane, wave);
#pragma unroll
    for (int c = 0; c < 16; ++c) Pnext[lane * BG + 16 * wave + c] = a[c];
}

// pivrow: 2 n doubles of scratch (scaled pivot row, then the pivot column)
int invert_spd_inplace(gl_ctx *ctx, double *A, int64_t n, double *pivrow) {
    if (n <= GJ_LDS_MAXN && !getenv("GRAPHLEARN_AMD_LGMRF_GJ_LAUNCH")) {
        k_gj_lds<<<1, 1024, 0, ctx->stream>>>(A, (int)n);
        GL_HIP(ctx, hipGetLastError());
        return GL_OK;
    }
    if (!getenv("GRAPHLEARN_AMD_LGMRF_GJ_LAUNCH")) {  // blocked
        const int nbk = (int)((n + BG - 1) / BG);
        double *scr = nullptr;
        GL_TRY(gl_ctx_scratch(ctx, ((size_t)2 * nbk * BG * BG + 2 * BG * BG) * 8, (void **)&scr));
        double *P[2] = {scr, scr + BG * BG}, *Rbuf = scr + 2 * BG * BG, *Cbuf = Rbuf + (size_t)nbk * BG * BG;
        k_bgj_pivot0<<<1, 256, 0, ctx->stream>>>(A, n, P[0]);
        for (int kb = 0; kb < nbk; ++kb) {
            k_bgj_panel<<<dim3((unsigned)nbk, 2), 1024, 0, ctx->stream>>>(A, n, kb, P[kb & 1], Rbuf, Cbuf);
            k_bgj_update<<<dim3((unsigned)nbk, (unsigned)nbk), 256, 0, ctx->stream>>>(A, n, kb, P[kb & 1], P[(kb + 1) & 1], Rbuf,
                                                                                      Cbuf, nbk);
        }
        GL_HIP(ctx, hipGetLastError());
        return GL_OK;
    }
    const dim3 g1((unsigned)((n + 255) / 256)), g2((unsigned)((n + 255) / 256), (unsigned)n);
    double *pivcol = pivrow + n;
    for (int64_t k = 0; k < n; ++k) {
        k_gj_pivot<<<g1, 256, 0, ctx->stream>>>(A, n, k, pivrow, pivcol);
        k_gj_eliminate<<<g2, 256, 0, ctx->stream>>>(A, n, k, pivrow, pivcol);
    }
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

int relative_error_dev(gl_ctx *ctx, const double *a_dev, const double *b_dev, int64_t count, double *part_dev,
                       double *out) {
    const int blocks = (int)std::min<int64_t>(256, (count + 255) / 256);
    k_relerr_part<<<blocks, 256, 0, ctx->stream>>>(a_dev, b_dev, count, part_dev);
    GL_HIP(ctx, hipGetLastError());
    std::vector<double> host((size_t)3 * blocks);
    GL_TRY(gl_d2h(ctx, host.data(), part_dev, host.size() * 8));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double d2 = 0, a2 = 0, am = 0;
    for (int b = 0; b < blocks; ++b) {
        d2 += host[3 * b];
        a2 += host[3 * b + 1];
        am = std::max(am, host[3 * b + 2]);
    }
    *out = (am <= 1e-8) ? sqrt(d2) : sqrt(d2) / sqrt(a2);  // evaluation.py:21-25
    return GL_OK;
}

}  // namespace

// relative_error(y_true, y_pred) of graph_learn/evaluation.py:8-25 on host buffers.
GL_API int gl_relative_error(gl_ctx *ctx, const double *y_true, const double *y_pred, int64_t count, double *out) {
    if (!ctx) return GL_E_INVALID;
    if (!y_true || !y_pred || !out || count < 1) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    Buf a, b, part;
    GL_MALLOC(ctx, &a.p, (size_t)count * 8, "lgmrf");
    GL_MALLOC(ctx, &b.p, (size_t)count * 8, "lgmrf");
    GL_MALLOC(ctx, &part.p, 3 * 256 * 8, "lgmrf");
    GL_TRY(gl_h2d(ctx, a.p, y_true, (size_t)count * 8));
    GL_TRY(gl_h2d(ctx, b.p, y_pred, (size_t)count * 8));
    return relative_error_dev(ctx, (double *)a.p, (double *)b.p, count, (double *)part.p, out);
}

// nonnegative_qp_solver(mat, vec, step, tol, max_iter) of graph_learn/lgmrf.py:19-71 on host buffers:
// explicit inverse of I + step*mat (Gauss-Jordan on the device), then ADMM.
GL_API int gl_nonneg_qp_solve(gl_ctx *ctx, const double *mat, const double *vec, int64_t m, double step, double tol,
                              int max_iter, double *z_out, int *iters_out) {
    if (!ctx) return GL_E_INVALID;
    if (!mat || !vec || !z_out || m < 1 || max_iter < 0) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    // one allocation for everything (seven hipMalloc / hipFree pairs were half of a small call's time)
    Buf arena;
    struct View {
        void *p = nullptr;
    } W, vecs, scal, part, bar, gran;
    {
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t bW = up((size_t)m * m * 8), bPart = up((size_t)4 * m * 6 * 8), bBar = up(GL_BAR_WORDS * 4),
                     bGran = up((size_t)2 * m * 16), bVec = up((size_t)16 * m * 8), bScal = up(16 * 8);
        GL_MALLOC(ctx, &arena.p, bW + bPart + bBar + bGran + bVec + bScal, "lgmrf");
        char *at = (char *)arena.p;
        W.p = at, at += bW;
        part.p = at, at += bPart;
        bar.p = at, at += bBar;       // bar, gran: zeroed together
        gran.p = at, at += bGran;
        vecs.p = at, at += bVec;
        scal.p = at;
        GL_HIP(ctx, hipMemsetAsync(bar.p, 0, bBar + bGran, ctx->stream));
    }
    AdmmRes ares;
    GL_TRY(admm_res_plan(ctx, m, ares));
    ares.bar = (unsigned *)bar.p;
    ares.gran = (lg_u4 *)gran.p;
    double *V = (double *)vecs.p;
    LgDev D{};
    D.n = m;
    D.W = (double *)W.p;
    D.p = V;
    D.x = V + m;
    D.z = V + 2 * m;
    D.uu = V + 3 * m;
    D.v = V + 4 * m;
    D.y = V + 5 * m;
    D.act = V + 6 * m;
    D.gcol = V + 7 * m;
    double *pivrow = V + 8 * m;
    D.scal = (double *)scal.p;
    D.part = (double *)part.p;
    D.rho = step;
    D.qp_tol = tol;
    // W = I + step * mat ; p = vec ; v0 = step * vec ; everything active
    std::vector<double> hw((size_t)m * m), hv((size_t)9 * m, 0.0);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < m; ++j) hw[(size_t)(i * m + j)] = (i == j ? 1.0 : 0.0) + step * mat[i * m + j];
    for (int64_t i = 0; i < m; ++i) {
        hv[(size_t)i] = vec[i];
        hv[(size_t)(4 * m + i)] = (0.0 - 0.0) + step * vec[i];
        hv[(size_t)(6 * m + i)] = 1.0;
    }
    GL_TRY(gl_h2d(ctx, D.W, hw.data(), hw.size() * 8));
    GL_TRY(gl_h2d(ctx, V, hv.data(), hv.size() * 8));
    GL_HIP(ctx, hipMemsetAsync(D.scal, 0, 16 * 8, ctx->stream));
    GL_TRY(invert_spd_inplace(ctx, D.W, m, pivrow));
    GL_TRY(ares.wave ? run_admm_wave(ctx, D, max_iter)
           : ares.quad ? run_admm_quad(ctx, D, max_iter)
                     : (ares.one ? run_admm_one(ctx, D, ares, max_iter)
                                 : (ares.on ? run_admm_res(ctx, D, ares, max_iter) : run_admm(ctx, D, max_iter))));
    double sc[9];
    GL_HIP(ctx, hipMemcpyAsync(sc, D.scal, 9 * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ares.on && sc[8] == 1.0) {  // the workgroups were not co-resident; nothing was touched: one launch per step instead
        GL_HIP(ctx, hipMemsetAsync(D.scal, 0, 16 * 8, ctx->stream));
        GL_TRY(run_admm(ctx, D, max_iter));
        GL_HIP(ctx, hipMemcpyAsync(sc, D.scal, 9 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (sc[8] != 0.0) return gl_set_error(ctx, GL_E_HIP, "ADMM: a hand-off between workgroups was lost in mid-run (GRAPHLEARN_AMD_LGMRF_ADMM=launch avoids the persistent forms)");
    GL_TRY(gl_d2h(ctx, z_out, D.z, (size_t)m * 8));
    if (iters_out) *iters_out = (int)sc[1];
    return GL_OK;
}

// LGMRF(norm_par, prob_tol, max_cycle, qp_step, qp_tol, regularization_type, adj_mask).fit(x)
// for the generalized Laplacian set (lgmrf.py:228-235).  x is n x n (not modified).
// path: 0 = automatic (fast when the mask is all ones / NULL), 1 = force general, 2 = force fast.
// probe_from / probe_to >= 0 (gl_lgmrf_time_node_updates only): ONE cycle is cut short after node update probe_to - 1 and
// the updates probe_from .. probe_to - 1 are bracketed by HIP events on the stream; the matrices are then NOT a fit and
// nothing is copied out.  The public fit always passes -1: no environment variable can truncate it.
static int lgmrf_fit_impl(gl_ctx *ctx, const double *x, int64_t n, double norm_par, double prob_tol, int max_cycle,
                          double qp_step, double qp_tol, int regularization_type, const double *adj_mask, int path,
                          double *laplacian_out, double *inv_laplacian_out, int *converged_out, int *cycles_out,
                          int64_t probe_from, int64_t probe_to, double *probe_seconds) {
    if (!ctx) return GL_E_INVALID;
    const bool probe = probe_from >= 0;
    if (!x || (!probe && (!laplacian_out || !inv_laplacian_out)) || n < 2) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    if (probe && (probe_to <= probe_from || probe_to > n || !probe_seconds)) return gl_set_error(ctx, GL_E_INVALID, "bad probe range");
    if (probe) max_cycle = 1;
    if (regularization_type != 1 && regularization_type != 2)
        return gl_set_error(ctx, GL_E_INVALID, "regularization type can be either 1 or 2, got %d", regularization_type);
    GL_HIP(ctx, hipSetDevice(ctx->device));

    bool full_mask = true;
    if (adj_mask)
        for (int64_t i = 0; i < n * n && full_mask; ++i) full_mask = adj_mask[i] == 1.0;
    if (path == 2 && !full_mask) return gl_set_error(ctx, GL_E_INVALID, "the fast path needs an all-ones adj_mask");
    const bool fast = (path == 2) || (path == 0 && full_mask);

    const size_t nn = (size_t)n * n * 8;
    Buf bx, bK, bTh, bC, bG, bW, bPrev, bMask, bVec, bScal, bPart;
    GL_MALLOC(ctx, &bx.p, nn, "lgmrf");
    GL_MALLOC(ctx, &bK.p, nn, "lgmrf");
    GL_MALLOC(ctx, &bTh.p, nn, "lgmrf");
    GL_MALLOC(ctx, &bC.p, nn, "lgmrf");
    GL_MALLOC(ctx, &bPrev.p, nn, "lgmrf");
    if (fast) GL_MALLOC(ctx, &bG.p, nn, "lgmrf");
    GL_MALLOC(ctx, &bW.p, nn, "lgmrf");
    if (adj_mask && !full_mask) {
        GL_MALLOC(ctx, &bMask.p, nn, "lgmrf");
        GL_TRY(gl_h2d(ctx, bMask.p, adj_mask, nn));
    }
    GL_MALLOC(ctx, &bVec.p, (size_t)16 * n * 8, "lgmrf");
    GL_MALLOC(ctx, &bScal.p, 16 * 8, "lgmrf");
    GL_MALLOC(ctx, &bPart.p, 3 * 256 * 8, "lgmrf");
    Buf bAdmm, bBar;
    Buf bGran, bCprev;
    GL_MALLOC(ctx, &bAdmm.p, (size_t)4 * n * 6 * 8, "lgmrf");  // one entry per workgroup and ring slot, any ADMM form
    GL_MALLOC(ctx, &bBar.p, GL_BAR_WORDS * 4, "lgmrf");
    GL_HIP(ctx, hipMemsetAsync(bBar.p, 0, GL_BAR_WORDS * 4, ctx->stream));
    GL_MALLOC(ctx, &bGran.p, (size_t)2 * n * 16, "lgmrf");
    GL_HIP(ctx, hipMemsetAsync(bGran.p, 0, (size_t)2 * n * 16, ctx->stream));
    AdmmRes ares;
    GL_TRY(admm_res_plan(ctx, n, ares));
    ares.bar = (unsigned *)bBar.p;
    ares.gran = (lg_u4 *)bGran.p;
    if (ares.on || !fast) GL_MALLOC(ctx, &bCprev.p, (size_t)n * n * 8, "lgmrf");
    GL_HIP(ctx, hipMemsetAsync(bVec.p, 0, (size_t)16 * n * 8, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(bScal.p, 0, 16 * 8, ctx->stream));

    LgDev D{};
    D.n = n;
    D.K = (double *)bK.p;
    D.Th = (double *)bTh.p;
    D.C = (double *)bC.p;
    D.G = (double *)bG.p;
    D.W = (double *)bW.p;
    D.ThPrev = (double *)bPrev.p;
    D.mask = (const double *)bMask.p;
    double *V = (double *)bVec.p;
    D.p = V;
    D.x = V + n;
    D.z = V + 2 * n;
    D.uu = V + 3 * n;
    D.v = V + 4 * n;
    D.y = V + 5 * n;
    D.act = V + 6 * n;
    D.beta = V + 7 * n;
    D.t = V + 8 * n;
    D.a = V + 9 * n;
    D.h = V + 10 * n;
    D.gcol = V + 11 * n;
    double *pivrow = V + 12 * n;
    D.scal = (double *)bScal.p;
    D.part = (double *)bAdmm.p;
    D.rho = qp_step;
    D.qp_tol = qp_tol;

    // x = x / np.mean(x) * norm_par   (lgmrf.py:229; the mean uses numpy's pairwise summation)
    const double mean = np_pairwise_sum(x, n * n) / (double)(n * n);
    GL_TRY(gl_h2d(ctx, bx.p, x, nn));
    const dim3 g2((unsigned)((n + 255) / 256), (unsigned)n);
    k_lg_prepare<<<g2, 256, 0, ctx->stream>>>((const double *)bx.p, D, mean, norm_par,
                                              regularization_type == 1 ? 2.0 : 1.0);
    GL_HIP(ctx, hipGetLastError());

    // general path: every node's active list nz(u) = {i != u : adj_mask[i, u] == 1} in ascending order (lgmrf.py:191-193),
    // known on the host once and for all; the device works on the gathered na x na problem (k_lg_compact_*)
    std::vector<int64_t> nz_ptr;
    Buf bIdx, bCvec;
    LgDev Dc = D;
    std::map<int64_t, AdmmRes> plans;  // the form of the ADMM loop by problem size na
    unsigned shared_tag = 0;           // the tagged forms share one record buffer: one tag sequence over all of them
    bool persistent_ok = true;         // false once a persistent form has found its workgroups not co-resident
    if (!fast) {
        std::vector<int> nz_idx;
        nz_ptr.assign((size_t)n + 1, 0);
        nz_idx.reserve((size_t)n * 8);
        for (int64_t u = 0; u < n; ++u) {
            for (int64_t i = 0; i < n; ++i)
                if (i != u && (!adj_mask || full_mask || adj_mask[i * n + u] == 1.0)) nz_idx.push_back((int)i);
            nz_ptr[(size_t)u + 1] = (int64_t)nz_idx.size();
        }
        GL_MALLOC(ctx, &bIdx.p, std::max<size_t>(nz_idx.size(), 1) * sizeof(int), "lgmrf");
        GL_MALLOC(ctx, &bCvec.p, (size_t)10 * n * 8, "lgmrf");
        GL_HIP(ctx, hipMemcpyAsync(bIdx.p, nz_idx.data(), nz_idx.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // nz_idx is a local
        double *Vc = (double *)bCvec.p;
        Dc.p = Vc, Dc.x = Vc + n, Dc.z = Vc + 2 * n, Dc.uu = Vc + 3 * n, Dc.v = Vc + 4 * n, Dc.y = Vc + 5 * n, Dc.act = Vc + 6 * n;
        Dc.gcol = Vc + 7 * n;
        Dc.G = nullptr;
        Dc.mask = nullptr;
        // problems of many sizes share the persistent kernels: their dynamic-LDS limits cover the largest plan there is
        const int lim = 150 * 1024, lim1 = (int)(((size_t)ADMM_ONE_MAXN * ADMM_ONE_MAXN + 2 * ADMM_ONE_MAXN + 6 * 16) * 8);
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_admm_tag<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_admm_tag<256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_admm_res, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_admm_one, hipFuncAttributeMaxDynamicSharedMemorySize, lim1));
    }
    const unsigned mvb = (unsigned)((n + 3) / 4);
    const int max_qp_iter = 100;  // default of nonnegative_qp_solver (lgmrf.py:24)
    int converged = -1, cycles = 0;
    // small problems on the fast path: the whole descent in one launch of one workgroup (k_lg_fit_small).
    // A forced ADMM form (GRAPHLEARN_AMD_LGMRF_ADMM) or GRAPHLEARN_AMD_LGMRF_FUSED=0 keeps the launch path.
    {
        const char *fz = getenv("GRAPHLEARN_AMD_LGMRF_FUSED");
        if (!probe && n <= 64 && !getenv("GRAPHLEARN_AMD_LGMRF_ADMM") && !(fz && !strcmp(fz, "0"))) {
            const size_t lds = ((size_t)4 * n * (n + 1) + 9 * n) * 8;
            double *res = D.scal + 12;
            const int np = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
            const bool general = !fast;
            if (general && !D.mask) {  // forced general path with a full mask: the kernel reads the mask
                GL_MALLOC(ctx, &bMask.p, nn, "lgmrf");
                std::vector<double> ones((size_t)n * n, 1.0);
                GL_TRY(gl_h2d(ctx, bMask.p, ones.data(), nn));
                GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                D.mask = (const double *)bMask.p;
            }
#define GL_FIT_SMALL(NPV, GEN)                                                                                              \
    do {                                                                                                                    \
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_lg_fit_small<NPV, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                         \
        k_lg_fit_small<NPV, GEN><<<1, 256, lds, ctx->stream>>>(D, max_cycle, prob_tol, max_qp_iter, res);                   \
    } while (0)
            if (np == 16 && !general) GL_FIT_SMALL(16, false);
            else if (np == 16) GL_FIT_SMALL(16, true);
            else if (np == 32 && !general) GL_FIT_SMALL(32, false);
            else if (np == 32) GL_FIT_SMALL(32, true);
            else if (!general) GL_FIT_SMALL(64, false);
            else GL_FIT_SMALL(64, true);
#undef GL_FIT_SMALL
            GL_HIP(ctx, hipGetLastError());
            double r2[2];
            GL_TRY(gl_d2h(ctx, laplacian_out, D.Th, nn));
            GL_TRY(gl_d2h(ctx, inv_laplacian_out, D.C, nn));
            GL_TRY(gl_d2h(ctx, r2, res, 2 * 8));
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (converged_out) *converged_out = (int)r2[0];
            if (cycles_out) *cycles_out = (int)r2[1];
            return GL_OK;
        }
    }
    // fused node update (fast path, tagged ADMM): K transposed into the buffer of the explicit inverse, which that path never
    // forms (a fit that falls back to one launch per step rebuilds W per node and no longer reads KT)
    const bool fuse3 = fast && ares.on && ares.tagged &&
                       !(getenv("GRAPHLEARN_AMD_LGMRF_FUSE3") && atoi(getenv("GRAPHLEARN_AMD_LGMRF_FUSE3")) == 0);
    bool gcol_ready = false;
    D.gcol_next = V + 14 * n;
    D.KT = nullptr;
    if (fuse3) {
        k_lg_transpose<<<dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(32, 8), 0, ctx->stream>>>(D.K, D.W, n);
        GL_HIP(ctx, hipGetLastError());
        D.KT = D.W;
    }
    for (int cycle = 0; cycle < max_cycle; ++cycle) {
        gcol_ready = false;
        GL_HIP(ctx, hipMemcpyAsync(D.ThPrev, D.Th, nn, hipMemcpyDeviceToDevice, ctx->stream));  // l_pre (:173)
        if (bCprev.p) GL_HIP(ctx, hipMemcpyAsync(bCprev.p, D.C, nn, hipMemcpyDeviceToDevice, ctx->stream));  // to redo the cycle
        if (fast) {  // fresh G = (Theta + rho I)^-1 once per cycle; rank-2 corrections in between
            k_lg_shift<<<g2, 256, 0, ctx->stream>>>(D.Th, D.G, n, D.rho);
            GL_TRY(invert_spd_inplace(ctx, D.G, n, pivrow));
        }
        const int64_t u_end = probe ? probe_to : n;
        for (int64_t u = 0; u < u_end; ++u) {
            if (probe && u == probe_from) GL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            bool fused_winv = false;
            if (!fast) {  // the gathered na x na problem of this node
                const int64_t na = nz_ptr[(size_t)u + 1] - nz_ptr[(size_t)u];
                const int *idx = (const int *)bIdx.p + nz_ptr[(size_t)u];
                Dc.n = na;
                k_lg_compact_vecs<<<1, VT, 0, ctx->stream>>>(D, Dc, u, idx);
                if (na > 0) {
                    k_lg_compact_w<<<dim3((unsigned)((na + 255) / 256), (unsigned)na), 256, 0, ctx->stream>>>(D, Dc, u, idx);
                    GL_TRY(invert_spd_inplace(ctx, Dc.W, na, pivrow));
                    auto it = plans.find(na);
                    if (it == plans.end()) {
                        AdmmRes pl;
                        GL_TRY(admm_res_plan(ctx, na, pl, false));  // (the LDS limits were raised once, below the lists)
                        pl.bar = ares.bar;
                        pl.gran = ares.gran;
                        it = plans.emplace(na, pl).first;
                    }
                    AdmmRes &pl = it->second;
                    if (pl.wave) GL_TRY(run_admm_wave(ctx, Dc, max_qp_iter));
                    else if (pl.quad) GL_TRY(run_admm_quad(ctx, Dc, max_qp_iter));
                    else if (pl.one) GL_TRY(run_admm_one(ctx, Dc, pl, max_qp_iter));
                    else if (pl.on && persistent_ok) {
                        pl.next_tag = shared_tag;
                        GL_TRY(run_admm_res(ctx, Dc, pl, max_qp_iter));
                        shared_tag = pl.next_tag;
                    } else GL_TRY(run_admm(ctx, Dc, max_qp_iter));
                }
                k_lg_scatter_beta<<<1, VT, 0, ctx->stream>>>(D, Dc, idx);
            } else if (fuse3 && ares.on && ares.tagged) {
                // Round 4: the node update in five launches instead of eight.  The QP's set-up rides in the ADMM kernel's
                // prologue (p = k_u / k_uu from the transposed K; column u of G is left in gcol by the previous node's
                // k_lg_update_g2, for the cycle's first node by k_lg_gcol), k_lg_node_update in k_lg_update_c2 and
                // k_lg_wood_scal in k_lg_update_g2: 69 -> 43 us of launches around the QP at n = 2048.
                if (u == 0 || !gcol_ready) k_lg_gcol<<<1, VT, 0, ctx->stream>>>(D, u);
                fused_winv = true;
                GL_TRY(run_admm_res(ctx, D, ares, max_qp_iter, u));
                k_matvec<<<mvb, 256, 0, ctx->stream>>>(D.C, D.beta, D.t, n, D.scal, 0);
                k_lg_update_c2<<<dim3((unsigned)((n + LG_RB - 1) / LG_RB)), VT, 0, ctx->stream>>>(D, u);
                k_matvec<<<mvb, 256, 0, ctx->stream>>>(D.G, D.a, D.h, n, D.scal, 0);
                k_lg_update_g2<<<dim3((unsigned)((n + LG_RB - 1) / LG_RB)), VT, 0, ctx->stream>>>(D, u, (u + 1) % n);
                std::swap(D.gcol, D.gcol_next);  // (u + 1's column of the corrected G)
                gcol_ready = u + 1 < n;          // (a new cycle starts from a fresh inverse)
                GL_HIP(ctx, hipGetLastError());
                continue;
            } else {
                k_lg_qp_setup<<<1, VT, 0, ctx->stream>>>(D, u);
                fused_winv = ares.wave || ares.quad || (ares.on && ares.tagged);  // the ADMM kernel forms its rows of the inverse itself
                if (!fused_winv) k_lg_build_winv<<<g2, 256, 0, ctx->stream>>>(D, u);
                GL_TRY(ares.wave ? run_admm_wave(ctx, D, max_qp_iter, fused_winv ? u : -1)
                       : ares.quad ? run_admm_quad(ctx, D, max_qp_iter, fused_winv ? u : -1)
                                 : (ares.one ? run_admm_one(ctx, D, ares, max_qp_iter)
                                             : (ares.on ? run_admm_res(ctx, D, ares, max_qp_iter, fused_winv ? u : -1)
                                                        : run_admm(ctx, D, max_qp_iter))));
                if (!fused_winv) k_lg_beta<<<1, VT, 0, ctx->stream>>>(D, u);  // else written by the ADMM kernel, with the snapshot
            }
            k_matvec<<<mvb, 256, 0, ctx->stream>>>(D.C, D.beta, D.t, n, D.scal, 0);
            if (!fused_winv) k_lg_snap_col<<<1, VT, 0, ctx->stream>>>(D, u);
            k_lg_node_update<<<1, VT, 0, ctx->stream>>>(D, u);
            k_lg_update_c<<<g2, 256, 0, ctx->stream>>>(D, u);
            if (fast) {
                k_matvec<<<mvb, 256, 0, ctx->stream>>>(D.G, D.a, D.h, n, D.scal, 0);
                k_lg_wood_scal<<<1, VT, 0, ctx->stream>>>(D, u);
                k_lg_update_g<<<g2, 256, 0, ctx->stream>>>(D);
            }
            GL_HIP(ctx, hipGetLastError());
        }
        cycles = cycle + 1;
        if (probe) {
            float ms = 0;
            GL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            GL_HIP(ctx, hipEventSynchronize(ctx->ev1));
            GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            *probe_seconds = (double)ms * 1e-3;
        }
        if (ares.on || (!fast && persistent_ok)) {  // error word of the persistent ADMM launches of this cycle
            double flag = 0;
            GL_HIP(ctx, hipMemcpyAsync(&flag, D.scal + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (flag == 1.0) {  // the workgroups were not co-resident (another process holds CUs): every later QP of the
                                // cycle has left at once; restore the cycle's start and redo it with one launch per step
                GL_HIP(ctx, hipMemcpyAsync(D.Th, D.ThPrev, nn, hipMemcpyDeviceToDevice, ctx->stream));
                GL_HIP(ctx, hipMemcpyAsync(D.C, bCprev.p, nn, hipMemcpyDeviceToDevice, ctx->stream));
                GL_HIP(ctx, hipMemsetAsync(D.scal, 0, 16 * 8, ctx->stream));
                ares.on = false;
                persistent_ok = false;
                D.KT = nullptr;  // (the launch form sets the QP up with k_lg_qp_setup and builds W, where K transposed lay)
                --cycle;
                continue;
            }
            if (flag != 0.0)
                return gl_set_error(ctx, GL_E_HIP, "LGMRF: a hand-off between the workgroups of the ADMM kernel was lost in "
                                                    "mid-run (GRAPHLEARN_AMD_LGMRF_ADMM=launch avoids the persistent forms)");
        }
        if (cycle > 4) {  // :221
            double rel = 0;
            GL_TRY(relative_error_dev(ctx, D.ThPrev, D.Th, n * n, (double *)bPart.p, &rel));
            if (rel < prob_tol) {
                converged = cycle;
                break;
            }
        } else {
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // bound the launch queue
        }
    }
    if (!probe) {
        GL_TRY(gl_d2h(ctx, laplacian_out, D.Th, nn));
        GL_TRY(gl_d2h(ctx, inv_laplacian_out, D.C, nn));
    }
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (converged_out) *converged_out = converged;
    if (cycles_out) *cycles_out = cycles;
    return GL_OK;
}

GL_API int gl_lgmrf_fit(gl_ctx *ctx, const double *x, int64_t n, double norm_par, double prob_tol, int max_cycle,
                        double qp_step, double qp_tol, int regularization_type, const double *adj_mask, int path,
                        double *laplacian_out, double *inv_laplacian_out, int *converged_out, int *cycles_out) {
    return lgmrf_fit_impl(ctx, x, n, norm_par, prob_tol, max_cycle, qp_step, qp_tol, regularization_type, adj_mask, path,
                          laplacian_out, inv_laplacian_out, converged_out, cycles_out, -1, -1, nullptr);
}

// Measurement only (bench.py's LGMRF leg, tools/): the node updates first_node .. first_node + count - 1 of the FIRST
// block-coordinate cycle (lgmrf.py:175-219), timed with HIP events on the context's stream.  Produces no matrices.
GL_API int gl_lgmrf_time_node_updates(gl_ctx *ctx, const double *x, int64_t n, double norm_par, double qp_step, double qp_tol,
                                      int regularization_type, const double *adj_mask, int path, int64_t first_node,
                                      int64_t count, double *seconds_out) {
    if (!ctx) return GL_E_INVALID;
    if (first_node < 0 || count < 1) return gl_set_error(ctx, GL_E_INVALID, "bad node range");
    return lgmrf_fit_impl(ctx, x, n, norm_par, 0.0, 1, qp_step, qp_tol, regularization_type, adj_mask, path, nullptr, nullptr,
                          nullptr, nullptr, first_node, first_node + count, seconds_out);
}
