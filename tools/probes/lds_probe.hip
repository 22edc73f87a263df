// Measures the LDS service time of 64-lane access patterns on gfx950 (ds_read_b128 / ds_write_b32), in wall-clock ticks
// (100 MHz) per 1000 instructions, for a table of candidate address patterns.  Test tooling for the layout of the
// backward's transposition buffer (render_bwd.hip); not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o lds_probe lds_probe.hip && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_read128(const int* __restrict__ addr, int ncand, int iters, unsigned long long* __restrict__ out, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    for (int c = 0; c < ncand; c++) {
        const uint32_t a = (uint32_t)addr[c * 64 + lane] * 4u;
        __syncthreads();
        unsigned long long t0 = wall_clock64();
        for (int it = 0; it < iters; it++) {
            f32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8 offset:16\n"
                "ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:16\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8 offset:16\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                : "v"(a));
            acc += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y + v6.z + v7.w;
        }
        __syncthreads();
        unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) out[c] = t1 - t0;
    }
    if (acc == 12345.f) sink[0] = acc;
}

__global__ void probe_write32(const int* __restrict__ addr, int ncand, int iters, unsigned long long* __restrict__ out, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < ncand; c++) {
        __syncthreads();
        const int a0 = addr[c * 64 + lane];
        const uint32_t a = (uint32_t)(a0 < 0 ? 0 : a0) * 4u;
        const bool on = a0 >= 0;
        unsigned long long t0 = wall_clock64();
        for (int it = 0; it < iters; it++) {
            const float v = (float)it;
            if (on)
                asm volatile(
                    "ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n"
                    "ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n"
                    "s_waitcnt lgkmcnt(0)\n" ::"v"(a), "v"(v) : "memory");
        }
        __syncthreads();
        unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) out[c] = t1 - t0;
    }
    if (lds[lane] == 12345.f) sink[0] = 1.f;
}

struct Cand { std::string name; int a[64]; };

int main() {
    std::vector<Cand> cands;
    auto add = [&](const std::string& name, auto f) { Cand c; c.name = name; for (int l = 0; l < 64; l++) c.a[l] = f(l); cands.push_back(c); };
    add("ideal: lane*4", [](int l) { return l * 4; });
    add("worst: lane*64", [](int l) { return l * 64; });
    add("stride 8 floats", [](int l) { return l * 8; });
    // round-3 split-bf16 flush: lane (kq, mm) reads row mm at 32 c + 8 kq, row stride 80 + 4 for rows 8..15
    add("mode0 r*80+4*(r>>3) c0", [](int l) { int kq = l >> 4, mm = l & 15; return mm * 80 + 4 * ((mm >> 3) & 1) + 8 * kq; });
    add("mode0 plain stride 68", [](int l) { int kq = l >> 4, mm = l & 15; return mm * 68 + 8 * kq; });
    add("mode0 plain stride 64", [](int l) { int kq = l >> 4, mm = l & 15; return mm * 64 + 8 * kq; });
    add("mode0 plain stride 72", [](int l) { int kq = l >> 4, mm = l & 15; return mm * 72 + 8 * kq; });
    const size_t fixed = cands.size();
    // split-f16 flush candidates: member row m = 2 (mm >> 2) + (mm & 1), chunk p = (mm >> 1) & 1; row(m) = m S + o1 (m&1) + o2 ((m>>1)&1) + o3 ((m>>2)&1)
    struct P { int S, o1, o2, o3; };
    std::vector<P> ps;
    for (int S = 64; S <= 88; S += 4)
        for (int o1 = 0; o1 <= 32; o1 += 4)
            for (int o2 = 0; o2 <= 32; o2 += 4)
                for (int o3 = 0; o3 <= 32; o3 += 4) {
                    int st[8];
                    for (int m = 0; m < 8; m++) st[m] = m * S + o1 * (m & 1) + o2 * ((m >> 1) & 1) + o3 * ((m >> 2) & 1);
                    std::sort(st, st + 8);
                    bool ok = true;
                    for (int m = 0; m < 7; m++) ok &= st[m + 1] - st[m] >= 64;
                    if (!ok || st[7] + 64 > 640) continue;
                    ps.push_back({S, o1, o2, o3});
                }
    for (auto& p : ps) {
        char nm[96];
        snprintf(nm, sizeof nm, "f16 W S=%d o=%d,%d,%d", p.S, p.o1, p.o2, p.o3);
        add(nm, [&](int l) { int kq = l >> 4, mm = l & 15, m = 2 * (mm >> 2) + (mm & 1), pl = (mm >> 1) & 1; return m * p.S + p.o1 * (m & 1) + p.o2 * ((m >> 1) & 1) + p.o3 * ((m >> 2) & 1) + 32 * pl + 8 * kq; });
        snprintf(nm, sizeof nm, "f16 H S=%d o=%d,%d,%d", p.S, p.o1, p.o2, p.o3);
        add(nm, [&](int l) { int kq = l >> 4, mm = l & 15, m = 2 * (mm >> 2) + (mm & 1), pl = (mm >> 1) & 1; return m * p.S + p.o1 * (m & 1) + p.o2 * ((m >> 1) & 1) + p.o3 * ((m >> 2) & 1) + 32 * (1 - pl) + 8 * kq; });
    }
    const int n = (int)cands.size();
    std::vector<int> h(n * 64);
    for (int c = 0; c < n; c++) for (int l = 0; l < 64; l++) h[c * 64 + l] = cands[c].a[l];
    int* d_addr; unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_out, n * 8); hipMalloc(&d_sink, 4);
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 100;
    std::vector<unsigned long long> o(n);
    probe_read128<<<1, 1024, 40960>>>(d_addr, n, iters, d_out, d_sink);
    hipMemcpy(o.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    printf("ds_read_b128: ticks (100 MHz) per %d instructions of 16 waves\n", iters * 8);
    for (size_t c = 0; c < fixed; c++) printf("  %-36s %6llu\n", cands[c].name.c_str(), o[c]);
    // rank the f16 candidates by W + H
    std::vector<std::pair<unsigned long long, int>> rank;
    for (size_t i = 0; i < ps.size(); i++) rank.push_back({o[fixed + 2 * i] + o[fixed + 2 * i + 1], (int)i});
    std::sort(rank.begin(), rank.end());
    printf("f16 candidates: %zu; best / worst (W + H ticks)\n", ps.size());
    for (size_t i = 0; i < rank.size(); i++)
        if (i < 12 || i + 3 >= rank.size() || (ps[rank[i].second].S == 72 && ps[rank[i].second].o1 == 4 && ps[rank[i].second].o2 == 0 && ps[rank[i].second].o3 == 0))
            printf("  #%zu S=%d o=%d,%d,%d  W %llu H %llu\n", i, ps[rank[i].second].S, ps[rank[i].second].o1, ps[rank[i].second].o2, ps[rank[i].second].o3,
                   o[fixed + 2 * rank[i].second], o[fixed + 2 * rank[i].second + 1]);
    // stores: staging write (64 consecutive floats), exchange-area patterns
    std::vector<Cand> w;
    auto addw = [&](const std::string& name, auto f) { Cand c; c.name = name; for (int l = 0; l < 64; l++) c.a[l] = f(l); w.push_back(c); };
    addw("write: lane", [](int l) { return l; });
    addw("write: exch mm<8 rho*8+mm", [](int l) { int kq = l >> 4, mm = l & 15; return mm < 8 ? (4 * kq) * 8 + mm : -1; });
    addw("write: exch mm<8 xrow(rho)+mm", [](int l) { int kq = l >> 4, mm = l & 15, rho = 4 * kq; return mm < 8 ? rho * 8 + 16 * (rho >> 3) + mm : -1; });
    addw("write: mode0 exch rows>=8", [](int l) { int kq = l >> 4, mm = l & 15; return (kq >= 2 && mm >= 4 && mm < 12) ? (4 * kq - 8) * 8 + mm - 4 : -1; });
    std::vector<int> hw(w.size() * 64);
    for (size_t c = 0; c < w.size(); c++) for (int l = 0; l < 64; l++) hw[c * 64 + l] = w[c].a[l];
    hipMemcpy(d_addr, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    probe_write32<<<1, 1024, 40960>>>(d_addr, (int)w.size(), iters, d_out, d_sink);
    hipMemcpy(o.data(), d_out, w.size() * 8, hipMemcpyDeviceToHost);
    printf("ds_write_b32: ticks per %d instructions of 16 waves\n", iters * 8);
    for (size_t c = 0; c < w.size(); c++) printf("  %-36s %6llu\n", w[c].name.c_str(), o[c]);
    return 0;
}
