// Write bandwidth of gfx950 as a function of the number of concurrent row streams (tuning probe, not part of the product): what the rulebook
// build does.  rulebook_kernel writes the table nbr[27][n] tap-major: a thread owns four consecutive rows and stores one int4 per tap, i.e. a
// wave writes 1 KB contiguous into each of the S tap rows it serves (S = 9 per workgroup: the ky split), the tap rows n * 4 bytes apart.
// Here the same store pattern with nothing else in the kernel: total bytes fixed (27 x n x 4), S streams per thread in {1, 3, 9, 27}
// (S = 1: the table as one sequential stream -- the row-major layout a [n][32] table would have is the S = 1 case with 128-byte rows).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/write_streams_probe tools/probes/write_streams_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int S>
__global__ void __launch_bounds__(256) writer(int *__restrict__ nbr, long long n, long long stride) {
    // grid.y = 27 / S groups of S taps; grid-stride over groups of four rows
    const int g = blockIdx.y;
    for (long long o4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; o4 < n; o4 += (long long)gridDim.x * blockDim.x * 4) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            int *dst = nbr + (long long)(g * S + k) * stride + o4;
            *reinterpret_cast<int4 *>(dst) = make_int4((int)o4, k, g, -1);
        }
    }
}

// the row-major alternative: a thread writes one row of 32 ints (27 taps + 5 pad) as eight int4 -> one sequential stream of 128-byte rows
__global__ void __launch_bounds__(256) writer_rowmajor(int *__restrict__ tab, long long n) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n * 8; o += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<int4 *>(tab)[o] = make_int4((int)o, 1, 2, -1);  // lane = (row, 16-byte piece): fully coalesced
}

int main() {
    const long long n = 543232;  // level-1 rows of two 317k-point clouds
    int *buf;
    hipMalloc(&buf, (size_t)32 * n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *what, auto launch, double bytes) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 1 && ms < best) best = ms;
        }
        printf("%-62s %7.1f us  %6.0f GB/s\n", what, best * 1e3f, bytes / (best * 1e-3) / 1e9);
    };
    const double bytes = 27.0 * n * 4;
    const unsigned bx = 256 * 32 / 3;  // rulebook_kernel's grid cap per ky
    run("27 x n x 4 B tap-major, 1 stream per thread (27 groups)", [&] { writer<1><<<dim3(bx, 27), 256>>>(buf, n, n); }, bytes);
    run("tap-major, 3 streams per thread (9 groups)", [&] { writer<3><<<dim3(bx, 9), 256>>>(buf, n, n); }, bytes);
    run("tap-major, 9 streams per thread (3 groups) [= rulebook_kernel]", [&] { writer<9><<<dim3(bx, 3), 256>>>(buf, n, n); }, bytes);
    run("tap-major, 27 streams per thread (1 group)", [&] { writer<27><<<dim3(bx, 1), 256>>>(buf, n, n); }, bytes);
    run("row-major [n][32] (128-byte rows, one sequential stream)", [&] { writer_rowmajor<<<dim3(8192), 256>>>(buf, n); }, 32.0 * n * 4);
    return 0;
}
