// Memory-side atomic throughput of gfx950 for the voxelizer's access pattern (tuning probe, not part of the product): the measured floor
// under vox_hash.  N threads, one "point" each; a point does what vox_hash does to its voxel's 16-byte hash entry
//   mode 0  nothing but the slot computation + the two result stores (the kernel's non-atomic skeleton)
//   mode 1  one 32-bit atomicCAS on entry.key (claim / find)
//   mode 2  mode 1 + one 64-bit atomicCAS on entry.fc (first point, count) -- what vox_hash issues per point
//   mode 3  mode 2 with the points of a wave that share a cell aggregated first (readfirstlane / ballot match loop): one pair of atomics per
//           (wave, cell), the members' arrival ranks from the leader's old count + their position in the match mask
// with V distinct cells drawn so that a wave of 64 consecutive points holds ~53 distinct ones and a cell ~2.2 points (the synthetic
// 10-sweep cloud: 317k points, 275k in range, 123k voxels, profiles/round6_voxelizer_floor.txt), table of 2^19 entries (8 MB).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/atomic_probe tools/probes/atomic_probe.hip ; run: tools/probes/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct __attribute__((aligned(16))) Entry {
    int key;
    int aux;
    unsigned long long fc;
};
constexpr unsigned long long kFcInit = 0x7fffffffull << 32;

__global__ void init(Entry *t, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { t[i].key = -1; t[i].aux = 0; t[i].fc = kFcInit; }
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(const int *__restrict__ keys, int n, Entry *__restrict__ table, unsigned mask, int *__restrict__ pslot, int *__restrict__ prank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int key = keys[i];
    unsigned slot = ((unsigned)key * 2654435761u >> 7) & mask;
    int rank = 0;
    if constexpr (MODE == 1 || MODE == 2) {
        bool fresh;
        while (true) {
            const int prev = atomicCAS(&table[slot].key, -1, key);
            fresh = prev == -1;
            if (fresh || prev == key) break;
            slot = (slot + 1) & mask;
        }
        if constexpr (MODE == 2) {
            unsigned long long *fc = &table[slot].fc;
            unsigned long long cur = fresh ? kFcInit : __hip_atomic_load(fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (true) {
                const unsigned first = (unsigned)(cur >> 32);
                const unsigned long long nw = ((unsigned long long)(first < (unsigned)i ? first : (unsigned)i) << 32) | (unsigned)((unsigned)cur + 1u);
                const unsigned long long prev = atomicCAS(fc, cur, nw);
                if (prev == cur) break;
                cur = prev;
            }
            rank = (int)(unsigned)cur;
        }
    }
    if constexpr (MODE == 3) {
        // match loop: the first remaining lane's key is broadcast, its group leaves together
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(1);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int k = __builtin_amdgcn_readlane(key, leader);
            const unsigned long long grp = __ballot(key == k) & todo;
            if (key == k) {
                const int members = __popcll(grp), pos = __popcll(grp & ((1ull << lane) - 1ull));
                unsigned long long cur = 0;
                if (lane == leader) {
                    bool fresh;
                    while (true) {
                        const int prev = atomicCAS(&table[slot].key, -1, key);
                        fresh = prev == -1;
                        if (fresh || prev == key) break;
                        slot = (slot + 1) & mask;
                    }
                    unsigned long long *fc = &table[slot].fc;
                    cur = fresh ? kFcInit : __hip_atomic_load(fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (true) {
                        const unsigned first = (unsigned)(cur >> 32);
                        const unsigned long long nw = ((unsigned long long)(first < (unsigned)i ? first : (unsigned)i) << 32) | (unsigned)((unsigned)cur + (unsigned)members);
                        const unsigned long long prev = atomicCAS(fc, cur, nw);
                        if (prev == cur) break;
                        cur = prev;
                    }
                }
                slot = (unsigned)__builtin_amdgcn_readlane((int)slot, leader);
                rank = __builtin_amdgcn_readlane((int)(unsigned)cur, leader) + pos;
            }
            todo &= ~grp;
        }
    }
    pslot[i] = (int)slot;
    prank[i] = rank;
}

int main() {
    const int n = 275000, n_cells = 123000;
    const unsigned slots = 1u << 19;
    std::vector<int> keys(n);
    srand(1);
    // consecutive groups of 64 points draw from a local pool of ~53 cells, pools reused across "sweeps" so that a cell gets ~2.2 points
    for (int i = 0; i < n; ++i) {
        const int wave = i / 64, sweep_span = n_cells / 53;
        const int pool = wave % sweep_span;
        keys[i] = (pool * 53 + rand() % 53) * 7 + 11;
    }
    int *d_keys, *d_slot, *d_rank;
    Entry *d_table;
    hipMalloc(&d_keys, n * 4); hipMalloc(&d_slot, n * 4); hipMalloc(&d_rank, n * 4); hipMalloc(&d_table, (size_t)slots * 16);
    hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            init<<<(slots + 255) / 256, 256>>>(d_table, slots);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            switch (mode) {
                case 0: probe<0><<<(n + 255) / 256, 256>>>(d_keys, n, d_table, slots - 1, d_slot, d_rank); break;
                case 1: probe<1><<<(n + 255) / 256, 256>>>(d_keys, n, d_table, slots - 1, d_slot, d_rank); break;
                case 2: probe<2><<<(n + 255) / 256, 256>>>(d_keys, n, d_table, slots - 1, d_slot, d_rank); break;
                default: probe<3><<<(n + 255) / 256, 256>>>(d_keys, n, d_table, slots - 1, d_slot, d_rank); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const char *what[] = {"no atomics (slot + two stores)", "key CAS", "key CAS + 64-bit (first, count) CAS  [= vox_hash]", "the same, wave-aggregated by a match loop"};
        printf("mode %d  %-55s %7.1f us for %d points  (%.1f G points/s)\n", mode, what[mode], best * 1e3f, n, n / (best * 1e-3f) / 1e9f);
    }
    return 0;
}
