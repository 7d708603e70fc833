// Host-side cost of the field / curve operations the MSM and Groth16 tails run on the CPU (no GPU needed).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I openzl_amd/csrc tools/host_ec_bench.hip -o tools/host_ec_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include "zl_ctx.h"
template <class F> static uint32_t low(const F& f) { uint32_t w; memcpy(&w, &f, 4); return w; }
template <class G>
static void run(const char* name) {
    using F = typename G::F;
    using clk = std::chrono::steady_clock;
    F x = G::gen_x(), y = G::gen_y();
    auto t0 = clk::now();
    F a = x;
    const int NM = 200000;
    for (int i = 0; i < NM; i++) a = zl::mul(a, y);
    auto t1 = clk::now();
    F s = x;
    for (int i = 0; i < NM; i++) s = zl::sqr(s);
    auto t2 = clk::now();
    XYZZ<F> p = XYZZ<F>::from_affine(Affine<F>{x, y});
    XYZZ<F> acc = p;
    const int ND = 20000;
    for (int i = 0; i < ND; i++) zl::dbl_inplace(acc);
    auto t3 = clk::now();
    XYZZ<F> acc2 = acc;
    for (int i = 0; i < ND; i++) zl::add_full(acc2, p);
    auto t4 = clk::now();
    XYZZ<F> acc3 = acc2;
    for (int i = 0; i < ND / 4; i++) zl::dbl_n(acc3, 4);
    auto t5 = clk::now();
    uint32_t k[8] = {0x12345678u, 0x9abcdef1u, 0x0fedcba9u, 0x87654321u, 0x13579bdfu, 0x2468ace0u, 0xdeadbeefu, 0x1234567u};
    XYZZ<F> r;
    const int NS = 50;
    for (int i = 0; i < NS; i++) { r = zl::mul_scalar_w4(acc3, k); k[0] += low(r.x); }
    auto t6 = clk::now();
    auto ns = [](auto a, auto b) { return std::chrono::duration<double, std::nano>(b - a).count(); };
    printf("%-10s mul %.1f ns  sqr %.1f ns  dbl %.1f ns  add_full %.1f ns  dbl_n(4)/4 %.1f ns  mul_scalar_w4 %.1f us  (%u %u %u)\n", name, ns(t0, t1) / NM,
           ns(t1, t2) / NM, ns(t2, t3) / ND, ns(t3, t4) / ND, ns(t4, t5) / ND, ns(t5, t6) / NS / 1000.0, low(a), low(s), low(r.x));
}
int main() {
    run<BlsG1>("bls g1");
    run<BnG1>("bn g1");
    run<BlsG2>("bls g2");
    return 0;
}
