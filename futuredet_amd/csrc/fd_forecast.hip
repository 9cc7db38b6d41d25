// Forecast association: the numeric core of `tracker` and `match_boxes`
// (det3d/datasets/nuscenes/nuscenes.py:112-123 match_boxes, :125-257 tracker, :84-98 box_*_center, :100-110 ... distance_matrix).
// The head emits, per sweep, up to 83 boxes for each of T forecast steps.  The reference chains them on the host with
// per-step numpy distance matrices; here one workgroup does all T-1 nearest-centre matchings forward and backward, walks
// the chains with the reject threshold, and extrapolates the constant-velocity trajectories, in float64 like numpy.
#include "fd_common.h"

namespace {

constexpr int kMaxT = 8;
constexpr int kMaxN = 256;

// distance_matrix (nuscenes.py:100-110): sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0)) with the products rounded as numpy does:
// (A*A).sum(axis=1) = fl(fl(a0*a0) + fl(a1*a1)); A.dot(B.T) is a dgemm -> fused multiply-add chain.
__device__ inline double dist2d(double a0, double a1, double b0, double b1) {
    const double ad = __dadd_rn(__dmul_rn(a0, a0), __dmul_rn(a1, a1));
    const double bd = __dadd_rn(__dmul_rn(b0, b0), __dmul_rn(b1, b1));
    const double dot = fma(a1, b1, __dmul_rn(a0, b0));
    double d = __dsub_rn(__dadd_rn(ad, bd), __dmul_rn(2.0, dot));
    if (d < 0.0) d = 0.0;
    return sqrt(d);
}

struct FcArgs {
    const double *centers;   // [T, n_max, 3]
    const double *velocity;  // [T, n_max, 3]
    const int *counts;       // [T]
    const double *time;      // [T-1]
    int T, n_max;
    double reject;
    int *fwd_idx, *fwd_ok, *bwd_idx, *bwd_ok, *match_idx, *status;
    double *cv_centers;
};

__global__ void __launch_bounds__(256) forecast_chains(FcArgs a) {
    __shared__ int s_idx[2][kMaxT - 1][kMaxN];
    __shared__ double s_dist[2][kMaxT - 1][kMaxN];
    __shared__ int s_cnt[kMaxT];
    __shared__ int s_empty;
    const int T = a.T, N = a.n_max, tid = threadIdx.x;
    if (tid < T) s_cnt[tid] = min(a.counts[tid], N);
    if (tid == 0) s_empty = 0;
    __syncthreads();
    if (tid < T && s_cnt[tid] == 0) s_empty = 1;  // tracker returns [] when any step is empty (nuscenes.py:150-158)
    // ---- all nearest-centre matchings: dir 0 = forward (curr + tm*v -> next), dir 1 = back-cast (curr - tm*v -> previous)
    for (int w = tid; w < 2 * (T - 1) * N; w += 256) {
        const int dir = w / ((T - 1) * N), r = w % ((T - 1) * N), s = r / N, i = r % N;
        // forward step s: current = t_s, other = t_{s+1}, tm = time[s]; backward step s: current = t_{T-1-s}, other = t_{T-2-s}, tm = time[T-2-s]
        const int tc = dir == 0 ? s : T - 1 - s, to = dir == 0 ? s + 1 : T - 2 - s;
        const double tm = a.time[dir == 0 ? s : T - 2 - s];
        int best = 0;
        double bd = 0.0;
        if (i < s_cnt[tc] && s_cnt[to] > 0) {
            const double *c = a.centers + ((size_t)tc * N + i) * 3, *v = a.velocity + ((size_t)tc * N + i) * 3;
            const double sgn = dir == 0 ? 1.0 : -1.0;
            const double p0 = __dadd_rn(c[0], __dmul_rn(sgn, __dmul_rn(tm, v[0])));  // center +- tm * velocity (nuscenes.py:88-98)
            const double p1 = __dadd_rn(c[1], __dmul_rn(sgn, __dmul_rn(tm, v[1])));
            bd = 1e300;
            for (int j = 0; j < s_cnt[to]; ++j) {
                const double *o = a.centers + ((size_t)to * N + j) * 3;
                const double d = dist2d(p0, p1, o[0], o[1]);
                if (d < bd) { bd = d; best = j; }  // np.argmin: first minimum
            }
        }
        s_idx[dir][s][i] = best;
        s_dist[dir][s][i] = bd;
    }
    // ---- match_boxes (nuscenes.py:112-123): every step's boxes re-ordered by nearest centre to the step-0 boxes
    for (int w = tid; w < T * N; w += 256) {
        const int t = w / N, i = w % N;
        int best = 0;
        if (i < s_cnt[0] && s_cnt[t] > 0) {
            const double *c = a.centers + (size_t)i * 3;
            double bd = 1e300;
            for (int j = 0; j < s_cnt[t]; ++j) {
                const double *o = a.centers + ((size_t)t * N + j) * 3;
                const double d = dist2d(c[0], c[1], o[0], o[1]);
                if (d < bd) { bd = d; best = j; }
            }
        }
        a.match_idx[w] = best;
    }
    __syncthreads();
    if (tid == 0) a.status[0] = s_empty;
    // ---- chains (nuscenes.py:160-173, 222-237): follow the matches, void when a hop is farther than the reject threshold
    for (int w = tid; w < 2 * N; w += 256) {
        const int dir = w / N, i = w % N;
        const int tstart = dir == 0 ? 0 : T - 1;
        int *out = (dir == 0 ? a.fwd_idx : a.bwd_idx) + (size_t)i * T;
        int ok = (!s_empty && i < s_cnt[tstart]) ? 1 : 0;
        int cur = i;
        out[0] = cur;
        for (int s = 0; s < T - 1; ++s) {
            if (ok || i < s_cnt[tstart]) {
                if (s_dist[dir][s][cur] > a.reject) ok = 0;
                cur = s_idx[dir][s][cur];
            }
            out[s + 1] = cur;
        }
        (dir == 0 ? a.fwd_ok : a.bwd_ok)[i] = ok;
    }
    // ---- constant velocity forward (nuscenes.py:183-193): center_{s+1} = center_s + time[s] * velocity(step-0 box), all 3 axes
    for (int w = tid; w < N * 3; w += 256) {
        const int i = w / 3, ax = w % 3;
        double c = a.centers[(size_t)i * 3 + ax];
        const double v = a.velocity[(size_t)i * 3 + ax];
        a.cv_centers[((size_t)i * T) * 3 + ax] = c;
        for (int s = 0; s < T - 1; ++s) {
            c = __dadd_rn(c, __dmul_rn(a.time[s], v));
            a.cv_centers[((size_t)i * T + s + 1) * 3 + ax] = c;
        }
    }
}

}  // namespace

extern "C" int fd_forecast_chains(const double *centers, const double *velocity, const int32_t *counts, const double *time_dev, int T,
                                  int n_max, double reject_thresh, int32_t *fwd_idx, int32_t *fwd_ok, int32_t *bwd_idx, int32_t *bwd_ok,
                                  int32_t *match_idx, double *cv_centers, int32_t *status, fd_stream_t stream) {
    FD_REQUIRE(centers && velocity && counts && time_dev && fwd_idx && fwd_ok && bwd_idx && bwd_ok && match_idx && cv_centers && status,
               "fd_forecast_chains: null argument");
    FD_REQUIRE(T >= 2 && T <= kMaxT, "fd_forecast_chains: T must be in [2,%d]", kMaxT);
    FD_REQUIRE(n_max >= 1 && n_max <= kMaxN, "fd_forecast_chains: n_max must be in [1,%d]", kMaxN);
    FcArgs a{centers, velocity, counts, time_dev, T, n_max, reject_thresh, fwd_idx, fwd_ok, bwd_idx, bwd_ok, match_idx, status, cv_centers};
    hipLaunchKernelGGL(forecast_chains, dim3(1), dim3(256), 0, fd::as_stream(stream), a);
    return fd::check_launch("fd_forecast_chains");
}
