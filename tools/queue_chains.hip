// Experiment (round 5): the cost of one step of a chain of tiny dependent kernels when S streams run such chains at once (one issuing host thread per stream).
// The 5-us kernels of a small proof's MSM tails and witness map show as 50-160 us in rocprofv3 timelines while three queues are active: is that the hardware
// (queues of one pipe taking turns) or the profiler?   hipcc --offload-arch=gfx950 -O2 -o /tmp/queue_chains tools/queue_chains.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_tiny(float* x, int spin) {
    float v = x[threadIdx.x];
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    x[threadIdx.x] = v;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 400, spin = argc > 2 ? atoi(argv[2]) : 200, blocks = argc > 3 ? atoi(argv[3]) : 1;
    for (int S : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(S);
        std::vector<float*> buf(S);
        for (int s = 0; s < S; s++) { (void)hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking); (void)hipMalloc(&buf[s], 256 * 4 * blocks); (void)hipMemset(buf[s], 0, 256 * 4 * blocks); }
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int s = 0; s < S; s++)
                th.emplace_back([&, s]() {
                    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, st[s], buf[s], spin);
                    (void)hipStreamSynchronize(st[s]);
                });
            for (auto& t : th) t.join();
            const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) best = dt < best ? dt : best;
        }
        printf("S=%d streams x %d dependent kernels (spin %d, %d blocks): %.2f us per chain step\n", S, N, spin, blocks, best / N);
        for (int s = 0; s < S; s++) { (void)hipStreamDestroy(st[s]); (void)hipFree(buf[s]); }
    }
    return 0;
}
