// Diagnostic build of oq_gemm.hip with in-kernel s_memrealtime stamps (MI355X_MICROARCH.md, DVFS item 6 / cdna guide section 7):
// where does the ~9 us per output tile go that a K sweep shows as the intercept of gemm_bf16_p3_kernel?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DOQ_GEMM_STAMPS tools/gemm_stamps.hip omniquant_amd/csrc/oq_api.cpp -o tools/gemm_stamps.bin
// Stamps per workgroup: 0 entry, 1 first K-tile landed (prologue barrier passed), 2 K loop done, 3 epilogue stores issued,
// 4 stores acknowledged.  Output: medians over workgroups of the phase lengths, the spread of entry / exit times, and the
// kernel's event duration.
#include "../omniquant_amd/csrc/oq_gemm.hip"
#include <algorithm>
#include <vector>

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    struct Shape { const char* name; int64_t M, N, K; int akc, bkc; };
    const Shape shapes[] = {{"fprop o     ", 2048, 4096, 4096, 1, 1}, {"fprop qkv   ", 2048, 12288, 4096, 1, 1},
                            {"fprop down  ", 2048, 4096, 11008, 1, 1}, {"dgrad o     ", 2048, 4096, 4096, 1, 0},
                            {"wgrad o     ", 4096, 4096, 2048, 0, 0}, {"wgrad qkv   ", 12288, 4096, 2048, 0, 0},
                            {"wgrad g|u   ", 22016, 4096, 2048, 0, 0}};
    const size_t maxel = (size_t)22016 * 4096;
    bf16_t *a, *b, *c;
    hipMalloc(&a, maxel * 2 * 2); hipMalloc(&b, maxel * 2 * 2); hipMalloc(&c, maxel * 2);
    std::vector<uint16_t> h(maxel * 2);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 9) & 0x3ff)) ^ (uint16_t)((x >> 3) & 0x8000); }  // +-[0.5, 2) bf16-ish
    hipMemcpy(a, h.data(), maxel * 2 * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), maxel * 2 * 2, hipMemcpyHostToDevice);
    const int maxwg = 4096;
    unsigned long long* st;
    hipMalloc(&st, (size_t)maxwg * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &st, sizeof(st));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Shape& s : shapes) {
        const int64_t lda = s.akc ? s.K : s.M, ldb = s.bkc ? s.K : s.N;
        auto run = [&]() {
            return oq_gemm(a, b, c, nullptr, nullptr, s.M, s.N, s.K, lda, ldb, s.N, s.akc, s.bkc, OQ_BF16, OQ_BF16, 1.f, 1, 1, 0, 0, 0, 0, 0,
                           0, 0, nullptr);
        };
        for (int i = 0; i < 20; ++i) if (run() != 0) { printf("oq_gemm failed: %s\n", oq_last_error()); return 1; }
        hipDeviceSynchronize();
        hipMemset(st, 0, (size_t)maxwg * 8 * 8);
        hipEventRecord(e0, nullptr);
        run();
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const int nwg = (int)(((s.M + 255) / 256) * ((s.N + 127) / 128));
        std::vector<unsigned long long> hs((size_t)nwg * 8);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < nwg; ++w) { tmin = std::min(tmin, hs[w * 8]); tmax = std::max(tmax, hs[w * 8 + 4]); }
        std::vector<double> pro, loop, epi, ack, entry, exit_, life;
        for (int w = 0; w < nwg; ++w) {
            const unsigned long long* t = &hs[w * 8];
            pro.push_back((t[1] - t[0]) * 0.01); loop.push_back((t[2] - t[1]) * 0.01); epi.push_back((t[3] - t[2]) * 0.01);
            ack.push_back((t[4] - t[3]) * 0.01); entry.push_back((t[0] - tmin) * 0.01); exit_.push_back((tmax - t[4]) * 0.01);
            life.push_back((t[4] - t[0]) * 0.01);
        }
        std::vector<double> e2 = entry; std::sort(e2.begin(), e2.end());
        const int first = std::min(nwg, 256);
        printf("%s M=%lld N=%lld K=%lld tiles %d: event %.1f us, in-kernel span %.1f us | per workgroup (median us): prologue %.2f, "
               "K loop %.2f (%.3f per K-tile), epilogue issue %.2f, store ack %.2f, lifetime %.2f | entry of the first %d workgroups: "
               "median +%.2f, last +%.2f us | idle after exit: median %.2f us\n",
               s.name, (long long)s.M, (long long)s.N, (long long)s.K, nwg, ms * 1e3, (tmax - tmin) * 0.01, med(pro), med(loop),
               med(loop) / (s.K / 64), med(epi), med(ack), med(life), first, e2[first / 2], e2[first - 1], med(exit_));
    }
    return 0;
}
