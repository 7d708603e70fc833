// tools/graph_chain.hip -- round 6 feasibility probe: a chain of N tiny dependent kernels issued launch by launch against the same chain as ONE hipGraphLaunch
// (what a small MSM job is: ~16 kernels of 5-20 us).  Prints host issue time and wall time per chain, for 1 and 4 chains on 4 streams issued by one thread.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/graph_chain tools/graph_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(unsigned* p, unsigned spin) {
    unsigned v = p[threadIdx.x];
    for (unsigned i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    p[threadIdx.x] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 16, S = 4, REPS = 200;
    unsigned* d[S];
    hipStream_t st[S];
    hipGraphExec_t ex[S];
    for (int s = 0; s < S; s++) {
        CK(hipMalloc(&d[s], 256 * 4));
        CK(hipMemset(d[s], 0, 256 * 4));
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, st[s], d[s], 2000u);
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int chains : {1, 4}) {
        for (int mode = 0; mode < 2; mode++) {
            double issue = 0, wall = 0;
            for (int r = 0; r < REPS + 10; r++) {
                CK(hipDeviceSynchronize());
                const double t0 = now_us();
                for (int s = 0; s < chains; s++) {
                    if (mode == 0) for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, st[s], d[s], 2000u);
                    else CK(hipGraphLaunch(ex[s], st[s]));
                }
                const double t1 = now_us();
                CK(hipDeviceSynchronize());
                const double t2 = now_us();
                if (r >= 10) { issue += t1 - t0; wall += t2 - t0; }
            }
            printf("%d chain(s) of %d kernels, %s: host issue %.1f us, wall %.1f us\n", chains, N, mode ? "hipGraphLaunch" : "launch by launch", issue / REPS, wall / REPS);
        }
    }
    return 0;
}
