// Is  q = y * R; r = fmaf(-q, D, y); q2 = fmaf(r, R, q)  (R = RN(1 / D), D = 0.3f) the correctly rounded y / D for every float y in
// [0, 18.1]?  (Markstein's two-FMA division.)  And: is (int)q2 == (int)(y / D) -- the bin torch.histc takes?
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void)
{
    const float D = 0.3f;
    const float R = 1.0f / D;
    uint32_t top;
    float lim = 18.1f;
    memcpy(&top, &lim, 4);
    uint64_t bad_q = 0, bad_bin = 0;
    for (uint32_t b = 0; b <= top; ++b) {
        float y;
        memcpy(&y, &b, 4);
        const float want = y / D;
        const float q = y * R;
        const float r = fmaf(-q, D, y);
        const float q2 = fmaf(r, R, q);
        if (q2 != want) {
            if (bad_q < 5) printf("q mismatch y=%a want=%a got=%a\n", y, want, q2);
            ++bad_q;
        }
        if ((int)q2 != (int)want) ++bad_bin;
    }
    printf("R = %a; values checked %u; quotient mismatches %llu; bin mismatches %llu\n", R, top + 1, (unsigned long long)bad_q,
           (unsigned long long)bad_bin);
    // ... and in the kernel's own terms, for EVERY distance d the range test lets through (bits 0 .. bits(0.3f)) and the one
    // value every other d is clamped to (bits(0.3f) + 1): k = (int)q2 of y = d * 60 is torch's bin min((int)(y / D), 59) -- never 60
    // inside the range, so the kernel needs no clamp -- and exactly 60 (the row nobody reads) for the clamp value
    uint32_t dtop;
    float three = 0.3f;
    memcpy(&dtop, &three, 4);
    uint64_t bad_d = 0;
    for (uint32_t b = 0; b <= dtop + 1; ++b) {
        float d;
        memcpy(&d, &b, 4);
        const float y = d * 60.0f;
        const float q = y * R;
        const float r = fmaf(-q, D, y);
        const int k = (int)fmaf(r, R, q);
        int want = (int)(y / D);
        if (want > 59) want = 59;
        if (b <= dtop ? k != want : k != 60) {
            if (bad_d < 5) printf("bin mismatch d=%a k=%d want=%d\n", d, k, b <= dtop ? want : 60);
            ++bad_d;
        }
    }
    printf("distances checked %u (0 .. 0.3f and the clamp value): mismatches %llu\n", dtop + 2, (unsigned long long)bad_d);
    return 0;
}
