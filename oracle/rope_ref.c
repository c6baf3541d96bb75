/* CPU restatement of the reference's 2-D RoPE loop (TEST INFRASTRUCTURE ONLY: the checker, never
 * the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * Follows rope_2d_cpu, /root/reference/src/model/encoder/backbone/croco/curope/curope.cpp:11-47:
 * tokens[B,N,H,D] in place; Q = D/4; for the y half (x = 0) and the x half (x = 1) of every head row,
 * the pair (u, v) = (t[q + x*2Q], t[q + Q + x*2Q]) is rotated by angle = fwd * pos / base^(q/Q).
 * Pinned against tests/golden/rope_*.pt (outputs of the reference's own PyTorch fallback
 * croco/pos_embed.py:112-159 and, when built, of its C++ CPU path in oracle/_ref).
 */
#include <math.h>
#include <stdint.h>

void rope2d_ref_f32(float* tok, const int64_t* pos, int B, int N, int H, int D, int64_t stride_b,
                    int64_t stride_n, float base, float fwd) {
    const int Q = D / 4;
    for (int b = 0; b < B; b++)
        for (int x = 0; x < 2; x++)
            for (int n = 0; n < N; n++) {
                const int p = (int)pos[((int64_t)b * N + n) * 2 + x];
                for (int h = 0; h < H; h++) {
                    float* row = tok + b * stride_b + n * stride_n + (int64_t)h * D + x * 2 * Q;
                    for (int q = 0; q < Q; q++) {
                        const float u = row[q], v = row[q + Q];
                        const float ang = fwd * p / powf(base, q / (float)Q);
                        const float c = cosf(ang), s = sinf(ang);
                        row[q] = u * c - v * s;
                        row[q + Q] = v * c + u * s;
                    }
                }
            }
}
