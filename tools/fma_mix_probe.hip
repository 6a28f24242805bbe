// Development tool: what does gfx950's v_fma_mixlo_f16 (fp32 x fp32 + 0 -> fp16 in one instruction,
// which the compiler selects for "(half)(a * b)") return, next to an fp32 multiply followed by a
// conversion?  Compares both on random operands and tests host-side models of the one-instruction
// result: one rounding of the exact product, round-toward-zero, the product of operands narrowed first.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/lib/fma_mix_probe.bin tools/fma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void both(const float* a, const float* b, float* two_step, float* one_insn, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p = a[i] * b[i];
    __asm__("" : "+v"(p));
    two_step[i] = static_cast<float>(static_cast<_Float16>(p));
    unsigned r;
    __asm__("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    _Float16 h;
    unsigned short lo = static_cast<unsigned short>(r & 0xFFFFu);
    memcpy(&h, &lo, 2);
    one_insn[i] = static_cast<float>(h);
}

static float h16(double x) { return static_cast<float>(static_cast<_Float16>(x)); }
static float rtz16(float x) {
    float r = static_cast<float>(static_cast<_Float16>(x));
    if (std::fabs(r) > std::fabs(x)) {
        _Float16 h = static_cast<_Float16>(r);
        uint16_t u; memcpy(&u, &h, 2); u -= 1; memcpy(&h, &u, 2);
        r = static_cast<float>(h);
    }
    return r;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), t(n), o(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        // a: an fp16-representable value around 0.02 (a moment), b: an fp32 factor such as 0.9 or a client weight
        a[i] = static_cast<float>(static_cast<_Float16>(0.05f * (static_cast<int>(rnd() % 20001) - 10000) / 10000.0f));
        b[i] = (i & 1) ? 0.9f : static_cast<float>((rnd() % 1000003) / 1000003.0);
    }
    float *da, *db, *dt, *dn;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dt, n * 4)); CK(hipMalloc(&dn, n * 4));
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    both<<<n / 256, 256>>>(da, db, dt, dn, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), dn, n * 4, hipMemcpyDeviceToHost));
    long diff = 0, host_two = 0, m_single = 0, m_rtz = 0, m_narrow = 0, m_rtz_exact = 0;
    for (int i = 0; i < n; ++i) {
        const float p32 = a[i] * b[i];
        const double exact = static_cast<double>(a[i]) * static_cast<double>(b[i]);   // 11 x 24 bits: exact in a double
        host_two += (h16(p32) == t[i]);
        diff += (t[i] != o[i]);
        m_single += (h16(exact) == o[i]);
        m_rtz += (rtz16(p32) == o[i]);
        m_rtz_exact += (rtz16(static_cast<float>(exact)) == o[i]);
        m_narrow += (h16(static_cast<double>(a[i]) * h16(b[i])) == o[i]);
    }
    printf("operands: %d pairs (fp16-representable a, fp32 b)\n", n);
    printf("multiply then convert on the device == the same two steps on the host: %ld of %d\n", host_two, n);
    printf("v_fma_mixlo_f16 differs from multiply then convert: %ld of %d (%.3f %%)\n", diff, n, 100.0 * diff / n);
    printf("v_fma_mixlo_f16 == one rounding of the exact product to fp16: %ld of %d\n", m_single, n);
    printf("v_fma_mixlo_f16 == fp32 product rounded toward zero to fp16:   %ld of %d\n", m_rtz, n);
    printf("v_fma_mixlo_f16 == product with b narrowed to fp16 first:      %ld of %d\n", m_narrow, n);
    int shown = 0;
    for (int i = 0; i < n && shown < 6; ++i)
        if (t[i] != o[i]) { printf("  a=%a b=%a  two steps %a  one instruction %a\n", a[i], b[i], t[i], o[i]); ++shown; }
    return 0;
}
