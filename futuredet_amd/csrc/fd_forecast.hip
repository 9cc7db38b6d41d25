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

// One workgroup per sweep (blockIdx.x = sample of a batch; every array of FcArgs is the first sample's, the others follow at the
// array's own size).
__global__ void __launch_bounds__(1024) forecast_chains(FcArgs a) {
    __shared__ int s_idx[2][kMaxT - 1][kMaxN];
    __shared__ double s_dist[2][kMaxT - 1][kMaxN];
    __shared__ int s_cnt[kMaxT];
    __shared__ int s_empty;
    const int T = a.T, N = a.n_max, tid = threadIdx.x;
    {
        const size_t b = blockIdx.x;
        a.centers += b * T * N * 3; a.velocity += b * T * N * 3; a.counts += b * T; a.time += b * (T - 1);
        a.fwd_idx += b * N * T; a.bwd_idx += b * N * T; a.fwd_ok += b * N; a.bwd_ok += b * N; a.match_idx += b * T * N;
        a.status += b; a.cv_centers += b * N * T * 3;
    }
    if (tid < T) s_cnt[tid] = min(a.counts[tid], N);
    if (tid == 0) s_empty = 0;
    __syncthreads();
    if (tid < T && s_cnt[tid] == 0) s_empty = 1;  // tracker returns [] when any step is empty (nuscenes.py:150-158)
    // ---- all nearest-centre matchings: dir 0 = forward (curr + tm*v -> next), dir 1 = back-cast (curr - tm*v -> previous)
    for (int w = tid; w < 2 * (T - 1) * N; w += (int)blockDim.x) {
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
    for (int w = tid; w < T * N; w += (int)blockDim.x) {
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
    for (int w = tid; w < 2 * N; w += (int)blockDim.x) {
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
    for (int w = tid; w < N * 3; w += (int)blockDim.x) {
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

// ---------------------------------------------------------------------------------------------- detections -> global boxes
// _second_det_to_nusc_box (det3d/datasets/nuscenes/nusc_common.py:167-189) followed by _lidar_nusc_box_to_global (:192-216)
// as arithmetic on arrays.  pyquaternion / nuScenes-devkit semantics restated (neither is in the reference tree):
//   Quaternion(axis=[0,0,1], radians=a) = (cos(a/2), 0, 0, sin(a/2)) in float64;
//   Quaternion.rotation_matrix = (Q(q) . Qbar(q)^T)[1:,1:] after normalising a non-unit q (|1 - |q|^2| >= 1e-14);
//   Box.rotate(q): center = R.center, orientation = q*orientation (= Q(q).o), velocity = R.velocity; Box.translate(t): center += t.
struct Rigid {  // one rotate + translate step; identity when !on
    double q[4], t[3];
    int on;
};
struct DetArgs {
    const float *box3d;  // [n, 9] (x,y,z,w,l,h,vx,vy,yaw)
    int n;
    Rigid step[2];       // lidar -> ego (calibrated_sensor), ego -> global (ego_pose)
    double *center, *quat, *velocity;  // [n,3], [n,4], [n,3]
    float *size;                        // [n,3]
};

__device__ inline void quat_matrix(const double *q, double R[3][3]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double Q[4][4] = {{w, -x, -y, -z}, {x, w, -z, y}, {y, z, w, -x}, {z, -y, x, w}};
    const double P[4][4] = {{w, -x, -y, -z}, {x, w, z, -y}, {y, -z, w, x}, {z, y, -x, w}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc = fma(Q[i + 1][k], P[j + 1][k], acc);
            R[i][j] = acc;
        }
}

// one box: head row b9 = (x,y,z,w,l,h,vx,vy,yaw) -> center / quat / velocity / size of box i of the output arrays
__device__ inline void det_to_global_one(const float *__restrict__ b, const Rigid (&step)[2], size_t i, double *__restrict__ center,
                                         double *__restrict__ quat, double *__restrict__ velocity, float *__restrict__ size) {
    const float yaw = __fsub_rn(-b[8], 1.5707963267948966f);  // float32: -box3d[:, -1] - np.pi / 2
    const double half = (double)yaw / 2.0;
    double o[4] = {cos(half), 0.0, 0.0, sin(half)};
    double c[3] = {(double)b[0], (double)b[1], (double)b[2]};
    double v[3] = {(double)b[6], (double)b[7], 0.0};
    for (int s = 0; s < 2; ++s) {
        if (!step[s].on) continue;
        double q[4] = {step[s].q[0], step[s].q[1], step[s].q[2], step[s].q[3]};
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (fabs(1.0 - n2) >= 1e-14 && n2 > 0.0) {
            const double nn = sqrt(n2);
            for (int k = 0; k < 4; ++k) q[k] /= nn;
        }
        double R[3][3];
        quat_matrix(q, R);
        double c2[3], v2[3];
        for (int r = 0; r < 3; ++r) {
            c2[r] = fma(R[r][2], c[2], fma(R[r][1], c[1], R[r][0] * c[0]));
            v2[r] = fma(R[r][2], v[2], fma(R[r][1], v[1], R[r][0] * v[0]));
        }
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        const double o2[4] = {w * o[0] - x * o[1] - y * o[2] - z * o[3], x * o[0] + w * o[1] - z * o[2] + y * o[3],
                              y * o[0] + z * o[1] + w * o[2] - x * o[3], z * o[0] - y * o[1] + x * o[2] + w * o[3]};
        for (int r = 0; r < 3; ++r) { c[r] = c2[r] + step[s].t[r]; v[r] = v2[r]; }
        for (int k = 0; k < 4; ++k) o[k] = o2[k];
    }
    for (int r = 0; r < 3; ++r) {
        center[i * 3 + r] = c[r];
        velocity[i * 3 + r] = v[r];
        size[i * 3 + r] = b[3 + r];
    }
    for (int k = 0; k < 4; ++k) quat[i * 4 + k] = o[k];
}

__global__ void __launch_bounds__(128) det_to_global_kernel(DetArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    det_to_global_one(a.box3d + (size_t)i * 9, a.step, (size_t)i, a.center, a.quat, a.velocity, a.size);
}

// The same per box, straight from the head's packed output [B, T, post, row_floats] (row = box 9 + score + label) with the two records
// of every sample in DEVICE memory ([B][14] = calibrated_sensor rotation wxyz, translation xyz, ego_pose rotation, translation; NULL:
// the boxes stay in the lidar frame): nothing of a sample is a kernel argument, so the launch can sit in a captured graph that is replayed
// for other samples.  Slots past a step's count are computed like any other row (their inputs are the decode's padding) and never read
// by the association.
__global__ void __launch_bounds__(128) det_to_global_packed_kernel(const float *__restrict__ packed, int row_floats, int rows_per_sample, int n_total,
                                                                   const double *__restrict__ records, double *__restrict__ center,
                                                                   double *__restrict__ quat, double *__restrict__ velocity, float *__restrict__ size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    Rigid step[2];
    const double *rec = records ? records + (size_t)(i / rows_per_sample) * 14 : nullptr;
    for (int s = 0; s < 2; ++s) {
        step[s].on = rec != nullptr;
        for (int k = 0; k < 4; ++k) step[s].q[k] = rec ? rec[7 * s + k] : (k == 0 ? 1.0 : 0.0);
        for (int k = 0; k < 3; ++k) step[s].t[k] = rec ? rec[7 * s + 4 + k] : 0.0;
    }
    det_to_global_one(packed + (size_t)i * row_floats, step, (size_t)i, center, quat, velocity, size);
}

// ---------------------------------------------------------------------------------------------- multi_future groups
// multi_future (nuscenes.py:299-339): boxes whose centres are closer than match_thresh are linked; network_split gives
// every connected component an id; networkx enumerates components in the order of their smallest member, so the id of a
// box is the rank of its component's smallest index.  One workgroup: min-label propagation over the adjacency
// (distance_matrix on all three coordinates, float64, numpy's rounding order), then the rank.
constexpr int kMaxGroupN = 8192;  // labels of one (sample, class) live in LDS; a thread owns boxes i, i + 1024, ...
__device__ inline double dist3d(const double *a, const double *b) {
    const double ad = __dadd_rn(__dadd_rn(__dmul_rn(a[0], a[0]), __dmul_rn(a[1], a[1])), __dmul_rn(a[2], a[2]));
    const double bd = __dadd_rn(__dadd_rn(__dmul_rn(b[0], b[0]), __dmul_rn(b[1], b[1])), __dmul_rn(b[2], b[2]));
    const double dot = fma(a[2], b[2], fma(a[1], b[1], __dmul_rn(a[0], b[0])));
    double d = __dsub_rn(__dadd_rn(ad, bd), __dmul_rn(2.0, dot));
    if (d < 0.0) d = 0.0;
    return sqrt(d);
}

__global__ void __launch_bounds__(1024) forecast_groups_kernel(const double *__restrict__ centers, int n, double thresh, int *__restrict__ ids) {
    __shared__ int s_label[kMaxGroupN];
    constexpr int kPer = kMaxGroupN / 1024;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) s_label[i] = i;
    __syncthreads();
    for (int it = 0; it < n; ++it) {  // a label travels at least one hop per sweep: at most n sweeps
        int best[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = tid + u * 1024;
            best[u] = i < n ? s_label[i] : 0;
            if (i < n)
                for (int j = 0; j < n; ++j)
                    if (s_label[j] < best[u] && dist3d(centers + (size_t)i * 3, centers + (size_t)j * 3) < thresh) best[u] = s_label[j];
        }
        __syncthreads();  // every read of this sweep is done before a label changes
        int changed = 0;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = tid + u * 1024;
            if (i < n && best[u] != s_label[i]) { s_label[i] = best[u]; changed = 1; }
        }
        // the vote IS the barrier: every thread leaves it with the same answer (the flag-in-LDS form let thread 0 reset the
        // flag for the next sweep while a slower wave had not read it yet)
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = tid; i < n; i += 1024) {
        const int lab = s_label[i];
        int rank = 0;
        for (int r = 0; r < lab; ++r) rank += (s_label[r] == r);
        ids[i] = rank;
    }
}

// ---------------------------------------------------------------------------------------------- trajectories of a sweep + their groups
// What `tracker` returns for one sweep, as arrays (nuscenes.py:160-241): the trajectories in the reference's order -- forward chains of the
// step-0 boxes that are not void, then the constant-velocity roll-out of EVERY step-0 box, then the back-cast chains of the last step's
// boxes that are not void (reversed, so they too start at step 0) -- and multi_future's forecast_id of each (nuscenes.py:299-339: connected
// components of "first boxes closer than match_thresh", numbered by their smallest member).  One workgroup per sweep.
//   kind  0 = forward chain, 1 = constant velocity, 2 = back-cast chain;  src = the box the trajectory was started from (step 0 for
//   kinds 0 / 1, last step for kind 2);  first = the step-0 box it begins with (kind 2: bwd_idx[src][T-1]).
constexpr int kMaxTraj = 3 * kMaxN;
__global__ void __launch_bounds__(1024) forecast_traj_groups_kernel(const double *__restrict__ centers, const int *__restrict__ counts, const int *__restrict__ fwd_ok,
                                                                    const int *__restrict__ bwd_ok, const int *__restrict__ bwd_idx, const int *__restrict__ status,
                                                                    int T, int N, double thresh, int *__restrict__ traj_kind, int *__restrict__ traj_src,
                                                                    int *__restrict__ traj_first, int *__restrict__ traj_group, int *__restrict__ n_traj) {
    __shared__ int s_label[kMaxTraj];
    __shared__ int s_first[kMaxTraj];
    __shared__ int s_scan[kMaxN + 1];
    __shared__ int s_n[3];
    extern __shared__ unsigned s_adj[];  // [n][ceil(n / 32)] adjacency bits, n <= 3 N
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x;
    centers += b * T * N * 3; counts += b * T; fwd_ok += b * N; bwd_ok += b * N; bwd_idx += b * N * T; status += b;
    traj_kind += b * 3 * N; traj_src += b * 3 * N; traj_first += b * 3 * N; traj_group += b * 3 * N; n_traj += b;
    const int n0 = status[0] ? 0 : min(counts[0], N), nl = status[0] ? 0 : min(counts[T - 1], N);
    // stable compaction of the two flag vectors (<= 256 entries each): thread 0 scans; the lists are tiny
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < n0; ++i) { s_scan[i] = run; run += fwd_ok[i] ? 1 : 0; }
        s_n[0] = run;
    }
    __syncthreads();
    const int nf = s_n[0];
    for (int i = tid; i < n0; i += 1024) {
        if (fwd_ok[i]) { const int p = s_scan[i]; traj_kind[p] = 0; traj_src[p] = i; s_first[p] = i; }
        const int p = nf + i;
        traj_kind[p] = 1; traj_src[p] = i; s_first[p] = i;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nl; ++i) { s_scan[i] = run; run += bwd_ok[i] ? 1 : 0; }
        s_n[1] = run;
    }
    __syncthreads();
    const int n = nf + n0 + s_n[1];
    for (int i = tid; i < nl; i += 1024)
        if (bwd_ok[i]) { const int p = nf + n0 + s_scan[i]; traj_kind[p] = 2; traj_src[p] = i; s_first[p] = bwd_idx[(size_t)i * T + T - 1]; }
    __syncthreads();
    if (tid == 0) n_traj[0] = n;
    for (int i = tid; i < n; i += 1024) { traj_first[i] = s_first[i]; s_label[i] = i; }
    for (int i = n + tid; i < 3 * N; i += 1024) { traj_kind[i] = -1; traj_src[i] = -1; traj_first[i] = -1; traj_group[i] = -1; }
    __syncthreads();
    // adjacency of the "closer than thresh" graph of the first boxes' centres (all three coordinates), one bit per ordered pair, computed ONCE
    // by all threads (n^2 float64 distances: the label sweeps below then only walk bits); then min-label propagation as in
    // forecast_groups_kernel
    const int nw = (n + 31) >> 5;
    for (int t = tid; t < n * nw; t += 1024) s_adj[t] = 0u;
    __syncthreads();
    for (int t = tid; t < n * n; t += 1024) {
        const int i = t / n, j = t - i * n;
        if (dist3d(centers + (size_t)s_first[i] * 3, centers + (size_t)s_first[j] * 3) < thresh) atomicOr(&s_adj[i * nw + (j >> 5)], 1u << (j & 31));
    }
    __syncthreads();
    for (int it = 0; it < n; ++it) {
        int best = tid < n ? s_label[tid] : 0;
        if (tid < n)
            for (int w = 0; w < nw; ++w) {
                unsigned m = s_adj[tid * nw + w];
                while (m) {
                    const int j = (w << 5) + __builtin_ctz(m);
                    m &= m - 1u;
                    best = min(best, s_label[j]);
                }
            }
        __syncthreads();
        int changed = 0;
        if (tid < n && best != s_label[tid]) { s_label[tid] = best; changed = 1; }
        if (!__syncthreads_or(changed)) break;
    }
    if (tid < n) {
        const int lab = s_label[tid];
        int rank = 0;
        for (int r = 0; r < lab; ++r) rank += (s_label[r] == r);
        traj_group[tid] = rank;
    }
}

// ---------------------------------------------------------------------------------------------- trajectory library lookup
// process_trajectories (nuscenes.py:341-382): every predicted trajectory, written as the row [vx, vy, q0..q3, centre_i - centre_0 ...],
// is replaced by the nearest row of a trajectory library (argmin over the library of the Euclidean distance, first minimum wins
// like np.argmin).  One workgroup per query row; float64 sums of squared differences in column order.
__global__ void __launch_bounds__(256) nearest_rows_kernel(const double *__restrict__ lib, int n_lib, const double *__restrict__ query, int dim,
                                                           int *__restrict__ idx) {
    __shared__ double s_d[256];
    __shared__ int s_i[256];
    const double *q = query + (size_t)blockIdx.x * dim;
    double best = 1.0e300;
    int arg = 0x7fffffff;
    for (int j = threadIdx.x; j < n_lib; j += 256) {
        const double *r = lib + (size_t)j * dim;
        double acc = 0.0;
        for (int c = 0; c < dim; ++c) {
            const double d = r[c] - q[c];
            acc = fma(d, d, acc);
        }
        if (acc < best) { best = acc; arg = j; }  // (j ascending per thread: the first minimum is kept)
    }
    s_d[threadIdx.x] = best;
    s_i[threadIdx.x] = arg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double d2 = s_d[threadIdx.x + o];
            const int i2 = s_i[threadIdx.x + o];
            if (d2 < s_d[threadIdx.x] || (d2 == s_d[threadIdx.x] && i2 < s_i[threadIdx.x])) { s_d[threadIdx.x] = d2; s_i[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) idx[blockIdx.x] = s_i[0];
}

}  // namespace

extern "C" int fd_nearest_rows(const double *library, int n_library, const double *queries, int n_queries, int dim, int32_t *idx, fd_stream_t stream) {
    FD_REQUIRE(n_queries >= 0 && n_library >= 1 && dim >= 1, "fd_nearest_rows: need n_library >= 1, dim >= 1");
    if (n_queries == 0) return FD_OK;
    FD_REQUIRE(library && queries && idx, "fd_nearest_rows: null argument");
    hipLaunchKernelGGL(nearest_rows_kernel, dim3((unsigned)n_queries), dim3(256), 0, fd::as_stream(stream), library, n_library, queries, dim, idx);
    return fd::check_launch("fd_nearest_rows");
}

extern "C" int fd_det_to_global_boxes(const float *box3d9, int n, const double *cs_rotation4, const double *cs_translation3,
                                      const double *pose_rotation4, const double *pose_translation3, double *center, double *quat,
                                      double *velocity, float *size, fd_stream_t stream) {
    FD_REQUIRE(n >= 0, "fd_det_to_global_boxes: negative count");
    if (n == 0) return FD_OK;
    FD_REQUIRE(box3d9 && center && quat && velocity && size, "fd_det_to_global_boxes: null argument");
    FD_REQUIRE((cs_rotation4 == nullptr) == (cs_translation3 == nullptr) && (pose_rotation4 == nullptr) == (pose_translation3 == nullptr),
               "fd_det_to_global_boxes: a rotation needs its translation");
    DetArgs a;
    a.box3d = box3d9; a.n = n; a.center = center; a.quat = quat; a.velocity = velocity; a.size = size;
    const double *rq[2] = {cs_rotation4, pose_rotation4}, *rt[2] = {cs_translation3, pose_translation3};
    for (int s = 0; s < 2; ++s) {
        a.step[s].on = rq[s] != nullptr;
        for (int k = 0; k < 4; ++k) a.step[s].q[k] = rq[s] ? rq[s][k] : (k == 0 ? 1.0 : 0.0);
        for (int k = 0; k < 3; ++k) a.step[s].t[k] = rt[s] ? rt[s][k] : 0.0;
    }
    hipLaunchKernelGGL(det_to_global_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, fd::as_stream(stream), a);
    return fd::check_launch("fd_det_to_global_boxes");
}

extern "C" int fd_forecast_groups(const double *centers3, int n, double match_thresh, int32_t *ids, fd_stream_t stream) {
    FD_REQUIRE(n >= 0 && n <= kMaxGroupN, "fd_forecast_groups: n must be in [0,%d]", kMaxGroupN);
    if (n == 0) return FD_OK;
    FD_REQUIRE(centers3 && ids, "fd_forecast_groups: null argument");
    hipLaunchKernelGGL(forecast_groups_kernel, dim3(1), dim3(1024), 0, fd::as_stream(stream), centers3, n, match_thresh, ids);
    return fd::check_launch("fd_forecast_groups");
}

extern "C" int fd_forecast_chains(const double *centers, const double *velocity, const int32_t *counts, const double *time_dev, int T,
                                  int n_max, double reject_thresh, int32_t *fwd_idx, int32_t *fwd_ok, int32_t *bwd_idx, int32_t *bwd_ok,
                                  int32_t *match_idx, double *cv_centers, int32_t *status, fd_stream_t stream) {
    FD_REQUIRE(centers && velocity && counts && time_dev && fwd_idx && fwd_ok && bwd_idx && bwd_ok && match_idx && cv_centers && status,
               "fd_forecast_chains: null argument");
    FD_REQUIRE(T >= 2 && T <= kMaxT, "fd_forecast_chains: T must be in [2,%d]", kMaxT);
    FD_REQUIRE(n_max >= 1 && n_max <= kMaxN, "fd_forecast_chains: n_max must be in [1,%d]", kMaxN);
    FcArgs a{centers, velocity, counts, time_dev, T, n_max, reject_thresh, fwd_idx, fwd_ok, bwd_idx, bwd_ok, match_idx, status, cv_centers};
    hipLaunchKernelGGL(forecast_chains, dim3(1), dim3(1024), 0, fd::as_stream(stream), a);
    return fd::check_launch("fd_forecast_chains");
}

// Head output of a batch -> global-frame boxes -> association -> trajectories and their groups: three launches, no host data of a sample
// among the kernel arguments (see include/futuredet_hip.h).
extern "C" int fd_forecast_from_detections(const float *packed, const int32_t *counts, int B, int T, int post, int row_floats, const double *records_dev,
                                           const double *time_dev, double reject_thresh, double match_thresh, const fd_forecast_buffers *out,
                                           fd_stream_t stream_) {
    FD_REQUIRE(B >= 0, "fd_forecast_from_detections: negative batch");
    if (B == 0) return FD_OK;
    FD_REQUIRE(packed && counts && time_dev && out, "fd_forecast_from_detections: null argument");
    FD_REQUIRE(T >= 2 && T <= kMaxT, "fd_forecast_from_detections: T must be in [2,%d]", kMaxT);
    FD_REQUIRE(post >= 1 && post <= kMaxN, "fd_forecast_from_detections: post must be in [1,%d]", kMaxN);
    FD_REQUIRE(row_floats >= 9, "fd_forecast_from_detections: a row holds at least the 9 box values");
    FD_REQUIRE(out->center && out->quat && out->velocity && out->size && out->fwd_idx && out->fwd_ok && out->bwd_idx && out->bwd_ok && out->match_idx &&
                   out->cv_centers && out->status,
               "fd_forecast_from_detections: null output buffer");
    const bool groups = out->traj_kind || out->traj_src || out->traj_first || out->traj_group || out->n_traj;
    FD_REQUIRE(!groups || (out->traj_kind && out->traj_src && out->traj_first && out->traj_group && out->n_traj),
               "fd_forecast_from_detections: the five trajectory buffers come together (or all NULL)");
    hipStream_t stream = fd::as_stream(stream_);
    const int rows = T * post, n_total = B * rows;
    hipLaunchKernelGGL(det_to_global_packed_kernel, dim3((unsigned)((n_total + 127) / 128)), dim3(128), 0, stream, packed, row_floats, rows, n_total, records_dev,
                       out->center, out->quat, out->velocity, out->size);
    FcArgs a{out->center, out->velocity, counts, time_dev, T, post, reject_thresh, out->fwd_idx, out->fwd_ok, out->bwd_idx, out->bwd_ok, out->match_idx,
             out->status, out->cv_centers};
    hipLaunchKernelGGL(forecast_chains, dim3((unsigned)B), dim3(1024), 0, stream, a);
    if (groups) {
        const size_t adj = (size_t)(3 * post) * ((3 * post + 31) / 32) * sizeof(unsigned);  // 8 KB at post = 83, 72 KB at the 256-box limit
        static std::atomic<uint64_t> lds_set{0};
        if (adj + 8 * 1024 > 65536 && !fd::ensure_dynamic_lds(reinterpret_cast<const void *>(forecast_traj_groups_kernel), adj, lds_set)) {
            fd::set_error("fd_forecast_from_detections: the runtime refused %zu bytes of LDS for the trajectory groups", adj);
            return FD_ELAUNCH;
        }
        hipLaunchKernelGGL(forecast_traj_groups_kernel, dim3((unsigned)B), dim3(1024), adj, stream, out->center, counts, out->fwd_ok, out->bwd_ok, out->bwd_idx,
                           out->status, T, post, match_thresh, out->traj_kind, out->traj_src, out->traj_first, out->traj_group, out->n_traj);
    }
    return fd::check_launch("fd_forecast_from_detections");
}
