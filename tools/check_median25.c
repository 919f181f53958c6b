/* Exhaustive 0-1-principle check of pysp_amd/csrc/median25_shared.inc + median25_rest.inc: a comparison network selects
 * the median of every input iff it does so for all 2^25 binary inputs.  gcc -O2 check_median25.c && ./a.out */
#include <stdint.h>
#include <stdio.h>
#define CE(a, b) { int lo = v[a] & v[b], hi = v[a] | v[b]; v[a] = lo; v[b] = hi; }
#define S3(a, b, c) { int lo = v[a] & v[b] & v[c], hi = v[a] | v[b] | v[c], md = (v[a] & v[b]) | (v[a] & v[c]) | (v[b] & v[c]); v[a] = lo; v[b] = md; v[c] = hi; }
static int med(int v[25]) {
#include "../pysp_amd/csrc/median25_shared.inc"
#include "../pysp_amd/csrc/median25_rest.inc"
    return v[12];
}
int main(void) {
    long bad = 0;
    for (uint32_t m = 0; m < (1u << 25); m++) {
        int v[25];
        for (int i = 0; i < 25; i++) v[i] = (m >> i) & 1;
        if (med(v) != (__builtin_popcount(m) >= 13)) bad++;
    }
    printf("median25 network: %ld failing binary inputs of %u\n", bad, 1u << 25);
    return bad != 0;
}
