// Two ways to let a side stream start a kernel once a kernel of the main stream has started or finished, and what
// each costs the main stream.  A chain of N kernels of ~20 us runs on the main stream; behind every kernel the side
// stream is released and runs one short kernel.
//   mode 0: no side stream (the chain alone)
//   mode 1: hipEventRecord on the main stream + hipStreamWaitEvent on the side stream (a fork per kernel)
//   mode 2: the NEXT chain kernel writes a flag as it starts; the side stream waits with hipStreamWaitValue32
//           (no packet on the main stream)
// Prints the chain time per kernel and the delay from the start of a chain kernel to the start of the side kernel
// it released.  Build: hipcc --offload-arch=gfx950 -O2 wait_value.hip -o wait_value
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__global__ void chain_kernel(unsigned* flag, unsigned val, long long* stamp, int spin) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (flag) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (stamp) *stamp = wall_clock64();
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
}
__global__ void side_kernel(long long* stamp) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *stamp = wall_clock64();
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const bool signal_mem = argc > 1 && atoi(argv[1]) != 0;     // 1: flags in hipMallocSignalMemory, 0: plain device memory
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d, flags in %s memory\n", can, signal_mem ? "signal" : "plain device");
    if (!can) return 0;
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 64;
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    hipEvent_t join; CK(hipEventCreateWithFlags(&join, hipEventDisableTiming | hipEventDisableSystemFence));
    unsigned* flags = nullptr;
    if (signal_mem) {
        // one 8-byte signal allocation per flag
        static unsigned* fl[N];
        for (int i = 0; i < N; ++i) CK(hipExtMallocWithFlags((void**)&fl[i], 8, hipMallocSignalMemory));
        flags = nullptr;
        long long* stamps; CK(hipMalloc(&stamps, 2 * N * sizeof(long long)));
        std::vector<long long> h(2 * N);
        const int spin = 2000;   // 100 MHz ticks: 20 us
        for (int rep = 0; rep < 3; ++rep) {
            const unsigned seq = 100 + rep;
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(chain_kernel, dim3(512), dim3(256), 0, m, fl[i], seq, stamps + i, spin);
                CK(hipStreamWaitValue32(s, fl[i], seq, hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(side_kernel, dim3(1), dim3(64), 0, s, stamps + N + i);
            }
            CK(hipEventRecord(join, s)); CK(hipStreamWaitEvent(m, join, 0));
            CK(hipStreamSynchronize(m));
            const double t1 = now();
            CK(hipMemcpy(h.data(), stamps, 2 * N * sizeof(long long), hipMemcpyDeviceToHost));
            double d = 0; for (int i = 0; i < N; ++i) d += (h[N + i] - h[i]) / 100.0;
            if (rep) printf("signal-memory flags: %.2f us per chain kernel, side start %.2f us after the chain kernel's start\n", (t1 - t0) / N, d / N);
        }
        return 0;
    }
    CK(hipMalloc(&flags, N * 256));
    CK(hipMemset(flags, 0, N * 256));
    long long* stamps; CK(hipMalloc(&stamps, 2 * N * sizeof(long long)));
    std::vector<long long> h(2 * N);
    const int spin = 2000;   // 100 MHz ticks: 20 us
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            const unsigned seq = 100 + rep + 10 * mode;
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < N; ++i) {
                unsigned* f = flags + i * 64;
                hipLaunchKernelGGL(chain_kernel, dim3(512), dim3(256), 0, m, mode == 2 ? f : nullptr, seq, stamps + i, spin);
                if (mode == 1) {
                    CK(hipEventRecord(ev[i], m)); CK(hipStreamWaitEvent(s, ev[i], 0));
                } else if (mode == 2) {
                    CK(hipStreamWaitValue32(s, f, seq, hipStreamWaitValueGte, 0xffffffffu));
                }
                if (mode) hipLaunchKernelGGL(side_kernel, dim3(1), dim3(64), 0, s, stamps + N + i);
            }
            if (mode) { CK(hipEventRecord(join, s)); CK(hipStreamWaitEvent(m, join, 0)); }
            CK(hipStreamSynchronize(m));
            const double t1 = now();
            CK(hipMemcpy(h.data(), stamps, 2 * N * sizeof(long long), hipMemcpyDeviceToHost));
            double d = 0; for (int i = 0; i < N; ++i) d += (h[N + i] - h[i]) / 100.0;
            double chain = (h[N - 1] - h[0]) / 100.0 / (N - 1);
            if (rep) printf("mode %d: wall %.2f us per chain kernel, start-to-start on the GPU %.2f us%s\n", mode, (t1 - t0) / N, chain,
                            mode ? (std::string(", side kernel starts ") + std::to_string(d / N) + " us after its chain kernel's start").c_str() : "");
        }
    return 0;
}
