// Probe of ds_read_b64_tr_b8 (gfx950): which bytes does lane i receive when lane j supplies address a_j?
//   hipcc -O2 --offload-arch=gfx950 tools/tr8_probe.hip -o tools/bin/tr8_probe && tools/bin/tr8_probe
// LDS image: byte at offset o holds (o & 0xFF) in plane 0 and (o >> 8) in plane 1 (two runs), so every delivered byte names its source offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* addr, uint32_t* out, int plane) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = plane ? (unsigned char)(i >> 8) : (unsigned char)(i & 0xFF);
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 2] = v[0];
    out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
    int* d_addr; uint32_t* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 2 * 64 * 2 * 4);
    // hypothesis A: lane l of a 16-lane group supplies row (l >> 1), 8 bytes at column 8 (l & 1) of a [rows][ROWB]-byte image
    for (int rowb : {16, 64}) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; addr[l] = g * 8 * rowb * 0 + 2048 * g + (i >> 1) * rowb + 8 * (i & 1); }
        hipMemcpy(d_addr, addr.data(), 64 * 4, hipMemcpyHostToDevice);
        std::vector<uint32_t> lo(128), hi(128);
        probe<<<1, 64>>>(d_addr, d_out, 0); hipMemcpy(lo.data(), d_out, 128 * 4, hipMemcpyDeviceToHost);
        probe<<<1, 64>>>(d_addr, d_out, 1); hipMemcpy(hi.data(), d_out, 128 * 4, hipMemcpyDeviceToHost);
        printf("row pitch %d bytes; lane: supplied offset -> offsets of its 8 received bytes\n", rowb);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr %5d :", l, addr[l]);
            for (int b = 0; b < 8; ++b) {
                const int o = ((hi[l * 2 + b / 4] >> (8 * (b % 4))) & 0xFF) * 256 + ((lo[l * 2 + b / 4] >> (8 * (b % 4))) & 0xFF);
                printf(" %5d", o);
            }
            printf("\n");
            if (l == 17) { printf("  ...\n"); l = 47; }
        }
    }
    return 0;
}
