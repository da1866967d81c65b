// Diagnostic: where wave 0 of each K1 workgroup spends its cycles (config-3 shape), plus the clock the chip
// holds (cycles of the median workgroup / wall time of the launch).
#define CAB_ATTN_STAMPS 1
#include "../cabinet_amd/csrc/cab_attn_fwd.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
int main() {
    const int B = 8, KC = 128, VC = 128, n = 1024;
    size_t nq = (size_t)B * KC * n;
    std::vector<float> h(nq);
    for (size_t i = 0; i < nq; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    float *q, *k, *v, *ctx, *lse;
    hipMalloc(&q, nq * 4); hipMalloc(&k, nq * 4); hipMalloc(&v, nq * 4); hipMalloc(&ctx, nq * 4); hipMalloc(&lse, B * n * 4);
    hipMemcpy(q, h.data(), nq * 4, hipMemcpyHostToDevice);
    hipMemcpy(k, h.data(), nq * 4, hipMemcpyHostToDevice);
    hipMemcpy(v, h.data(), nq * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0);
        cabinet::attn_fwd_dispatch(q, k, v, 0.088f, B, KC, VC, n, ctx, lse, nullptr, nullptr, 1, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    static unsigned long long st[4096][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(cabinet::cab_stamps), sizeof(st));
    const int nb = 256;
    auto med = [&](auto f) { std::vector<double> x; for (int b = 0; b < nb; ++b) x.push_back(f(b)); std::sort(x.begin(), x.end()); return x[nb / 2]; };
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nb; ++b) { tmin = std::min(tmin, st[b][0]); tmax = std::max(tmax, st[b][4]); }
    printf("launch (event, instrumented build) %.1f us; first entry -> last exit %llu ticks\n", ms * 1e3, tmax - tmin);
    printf("median per workgroup (wave 0), ticks:\n");
    printf("  prologue (entry -> first tile's S chain + softmax done) %8.0f\n", med([&](int b) { return (double)(st[b][1] - st[b][0]); }));
    printf("  pipelined loop (7 iterations)                           %8.0f   phase A sum %8.0f   phase B sum %8.0f\n",
           med([&](int b) { return (double)(st[b][2] - st[b][1]); }), med([&](int b) { return (double)st[b][5]; }), med([&](int b) { return (double)st[b][6]; }));
    printf("  last tile PV                                            %8.0f\n", med([&](int b) { return (double)(st[b][3] - st[b][2]); }));
    printf("  merge + store                                           %8.0f\n", med([&](int b) { return (double)(st[b][4] - st[b][3]); }));
    printf("  total                                                   %8.0f   (ideal MFMA 8*128*64 = 65536)\n", med([&](int b) { return (double)(st[b][4] - st[b][0]); }));
    printf("  start skew: max(entry) - min(entry) = %llu ticks\n", [&] { unsigned long long a = 0; for (int b = 0; b < nb; ++b) a = std::max(a, st[b][0]); return a - tmin; }());
    return 0;
}
