// Host cost of the HIP calls a training step is made of: kernel launch (small and 700-byte argument blocks), event
// record + stream wait pairs, memset.  Build: hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[704]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 12345) *p = b.b[0]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t ev[64]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    int* d; hipMalloc(&d, 1 << 20);
    Big big{};
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s0, d);
        double t1 = now(); hipDeviceSynchronize(); double t1d = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s0, big, d);
        double t2 = now(); hipDeviceSynchronize(); double t2d = now();
        for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], s0); hipStreamWaitEvent(s1, ev[i & 63], 0); }
        double t3 = now(); hipDeviceSynchronize(); double t3d = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s0, big, d); hipEventRecord(ev[i & 63], s0); hipStreamWaitEvent(s1, ev[i & 63], 0); hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s1, big, d); }
        double t4 = now(); hipDeviceSynchronize(); double t4d = now();
        for (int i = 0; i < N; ++i) hipMemsetAsync(d, 0, 4096, s0);
        double t5 = now(); hipDeviceSynchronize();
        if (rep) printf("launch small %.2f us (drain %.2f) | launch 704-B args %.2f us (drain %.2f) | record+wait %.2f us | launch+record+wait+launch %.2f us (drain %.2f) | memset %.2f us\n",
                        (t1 - t0) / N, (t1d - t0) / N, (t2 - t1d) / N, (t2d - t1d) / N, (t3 - t2d) / N, (t4 - t3d) / N, (t4d - t3d) / N, (t5 - t4d) / N);
    }
    return 0;
}
