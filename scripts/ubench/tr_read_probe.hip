// round 6 probe: what ds_read_b64_tr_b16 returns on gfx950, lane by lane.
// LDS holds lds16[i] = i.  Pattern A: lane l reads at byte address 8 l (64 consecutive 8-byte chunks).  Pattern B: the address the TN form of
// the split-plane GEMM would use on a [16 k][32 rows] piece (64-byte k-rows): k = 8 (l >> 5) + (l & 15) / 4, row = 16 ((l >> 4) & 1) + 4 (l & 3)
// -> expected result for lane l, element j: value at (k = 8 (l >> 5) + j, row = l & 31), i.e. 32 (8 (l >> 5) + j) + (l & 31).
// build: hipcc --offload-arch=gfx950 -O2 -o tr_read_probe tr_read_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint16_t* out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds16[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds16[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = 8 * l;
    else addr = (8 * (l >> 5) + ((l & 15) >> 2)) * 64 + ((l >> 4) & 1) * 32 + (l & 3) * 8;
    addr += (unsigned)(uintptr_t)lds16;      // LDS byte offset of the array (0 here; generic)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int pattern = 0; pattern < 2; ++pattern) {
        probe<<<1, 64>>>(d, pattern);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pattern);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (pattern == 1)
                for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != 32 * (8 * (l >> 5) + j) + (l & 31);
        }
        if (pattern == 1) printf("pattern 1 mismatches against the MFMA A-fragment expectation: %d\n", bad);
    }
    return 0;
}
