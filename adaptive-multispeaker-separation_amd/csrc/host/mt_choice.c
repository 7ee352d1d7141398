/* Host helper: the k-means restart seeds of the reference, bit-identical, without numpy's per-row cost.

   Reference: models/Kmeans_2.py:61-66 draws, for every row of the [B * nb_tries] k-means problems,
       np.random.choice(range(l), size=C, replace=False)
   from numpy's GLOBAL legacy generator (seeded 42 at models/network.py:17-18).  In numpy's legacy RandomState that call is
   permutation(l)[:C]: arange(l) shuffled by Fisher-Yates from the top,
       for i = l-1 .. 1:  j = interval(i);  swap(x[i], x[j])
   where interval(max) masks 32-bit MT19937 outputs with the smallest 2^k - 1 >= max and rejects values > max.  The draw is a
   property of the STREAM, so it has to consume exactly the words numpy would; what it does not have to do is the work numpy does
   around them (a Python-level call, an arange and l swaps per row).  Here:

     1. MT19937 blocks are regenerated and tempered 624 words at a time (plain loops the compiler vectorises: every read is
        >= 227 words away from the write);
     2. the rejection loop is branch-free: the candidate is stored unconditionally and i only moves when it is accepted; with
        AVX2, 32 candidates at a time are classified without knowing each other's outcome (accept_avx2);
     3. no array is shuffled: only x[0..C) of the result are wanted, and the final occupant of position p is found by walking the
        swap list BACKWARDS in time (i ascending): q = p; at swap (i, j_i): q == i -> j_i, q == j_i -> i.  After swap i the
        tracked position is <= i, so beyond i = C the only event is "j_i == q", a vectorised search.

   State in and out is numpy's own (key[624], pos) as returned by np.random.get_state() / accepted by set_state(), so other users
   of the global generator see the stream position numpy would have left.  Plain C, no dependencies; built into libams_host.so. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];       /* numpy's state->key */
    uint32_t out[MT_N];       /* the same block, tempered */
    int pos;                  /* numpy's state->pos: next word of the block, MT_N = block exhausted */
} mt_t;

/* The word stream as the rejection loops see it: the tempered words of the current block and the position in it. */
typedef struct {
    mt_t* gen;
    const uint32_t* w;        /* tempered words of the current block */
    int pos;                  /* next word of the current block */
} stream_t;

/* i - 1 + (i < v): the loop-carried step of the rejection loop.  On x86-64 as compare + add-with-carry (2 cycles); compilers turn the
   C form into setb / movzx / lea (5). */
static inline uint32_t step_down(uint32_t i, uint32_t v) {
#if defined(__x86_64__)
    __asm__("cmpl %[v], %[i]\n\tadcl $-1, %[i]" : [i] "+r"(i) : [v] "r"(v) : "cc");
    return i;
#else
    return i - 1 + (uint32_t)(i < v);
#endif
}

static inline uint32_t mt_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);                 /* numpy/randomkit: the low bit of y is the low bit of v */
}

__attribute__((target_clones("avx512f", "avx2", "default")))
static void mt_temper(mt_t* s) {
    for (int k = 0; k < MT_N; ++k) {
        uint32_t y = s->key[k];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        s->out[k] = y;
    }
}

__attribute__((target_clones("avx512f", "avx2", "default")))
static void mt_next_block(mt_t* s) {
    uint32_t* const mt = s->key;
    int k;
#pragma GCC ivdep
    for (k = 0; k < MT_N - MT_M; ++k) mt[k] = mt[k + MT_M] ^ mt_mix(mt[k], mt[k + 1]);
#pragma GCC ivdep
    for (; k < MT_N - 1; ++k) mt[k] = mt[k + (MT_M - MT_N)] ^ mt_mix(mt[k], mt[k + 1]);
    mt[MT_N - 1] = mt[MT_M - 1] ^ mt_mix(mt[MT_N - 1], mt[0]);
    mt_temper(s);
    s->pos = 0;
}

/* (Generating the blocks on a second thread was tried and removed: handing 5 KB per block from core to core costs more than
   the 0.2 ns/word the generation takes locally -- 15.6 ms against 8.2 ms per 640 rows on an EPYC 9575F.) */
static const uint32_t* stream_next_block(stream_t* st) {
    mt_next_block(st->gen);
    st->w = st->gen->out;
    st->pos = 0;
    return st->w;
}

/* largest k <= k0 with seq[k] == q, or -1 */
__attribute__((target_clones("avx512f", "avx2", "default")))
static int find_eq_down(const uint32_t* seq, int k0, uint32_t q) {
    int k = k0;
    for (; k >= 31; k -= 32) {
        uint32_t any = 0;
        for (int t = 0; t < 32; ++t) any |= (seq[k - t] == q);
        if (any) break;
    }
    for (; k >= 0; --k)
        if (seq[k] == q) return k;
    return -1;
}

/* C tracks at once, k descending from k0: a track whose position equals seq[k] moves to i = l - 1 - k. */
#define WALK_MAX 4
__attribute__((target_clones("avx512f", "avx2", "default")))
static void walk_down(const uint32_t* __restrict seq, int k0, int l, int C, uint32_t* __restrict q) {
    uint32_t t4[WALK_MAX];
    for (int p = 0; p < WALK_MAX; ++p) t4[p] = p < C ? q[p] : 0xfffffffeu;      /* unused tracks never match (values < 2^31) */
    int k = k0;
    while (k >= 31) {
        const uint32_t a = t4[0], b = t4[1], c = t4[2], d = t4[3];
        uint32_t any = 0;
        for (int t = 0; t < 32; ++t) {
            const uint32_t v = seq[k - t];
            any |= (uint32_t)(v == a) | (uint32_t)(v == b) | (uint32_t)(v == c) | (uint32_t)(v == d);
        }
        if (any) {
            for (int t = 0; t < 32; ++t)
                for (int p = 0; p < C; ++p)
                    if (seq[k - t] == t4[p]) t4[p] = (uint32_t)(l - 1 - (k - t));
        }
        k -= 32;
    }
    for (; k >= 0; --k)
        for (int p = 0; p < C; ++p)
            if (seq[k] == t4[p]) t4[p] = (uint32_t)(l - 1 - k);
    for (int p = 0; p < C; ++p) q[p] = t4[p];
}

/* The accepted draws of one row, in the order they are accepted: seq[n] = j_i for i = l - 1 - n.  Scalar form. */
static void accept_scalar(stream_t* s, int l, uint32_t* seq) {
    uint32_t i = (uint32_t)l - 1, mask = i;
    const uint32_t top = i;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    int pos = s->pos;
    while (i >= 1) {
        /* one mask level: i in (lo, mask].  Inside, the only loop-carried chain is step_down() on i; the candidate is stored
           unconditionally and i moves only when it is accepted (no data-dependent branch) */
        const uint32_t lo = mask >> 1;
        while (i > lo) {
            if (pos == MT_N) { stream_next_block(s); pos = 0; }
            const uint32_t* const w = s->w;
            /* chunks of 8 while neither bound can be crossed inside one (at most 8 acceptances, 8 words) */
            while (pos + 8 <= MT_N && i >= lo + 8) {
#pragma GCC unroll 8
                for (int k = 0; k < 8; ++k) {
                    const uint32_t v = w[pos + k] & mask;
                    seq[top - i] = v;
                    i = step_down(i, v);
                }
                pos += 8;
            }
            while (pos < MT_N && i > lo) {
                const uint32_t v = w[pos++] & mask;
                seq[top - i] = v;
                i = step_down(i, v);
            }
        }
        mask = lo;                                                   /* i == lo == 2^k - 1: the smallest mask >= i is lo itself */
    }
    s->pos = pos;
}

#if defined(__x86_64__)
#include <immintrin.h>
static uint32_t g_compress[256][8];      /* lane order that moves the lanes whose mask bit is set to the front, in order */
static int g_compress_ready = 0;
static void compress_init(void) {
    for (int m = 0; m < 256; ++m) {
        int n = 0;
        for (int b = 0; b < 8; ++b) if (m & (1 << b)) g_compress[m][n++] = (uint32_t)b;
        for (int b = 0; b < 8; ++b) if (!(m & (1 << b))) g_compress[m][n++] = (uint32_t)b;
    }
    g_compress_ready = 1;
}

/* AVX2 form.  With i far above the level's floor, a candidate v <= i - 32 is accepted and a candidate v > i is rejected WHATEVER
   happens to the 31 candidates around it (i moves by at most 32 inside a chunk of 32); only lo-probability values in (i - 32, i]
   depend on the order.  So 32 candidates are classified by two vector compares against constants of the chunk, the accepted ones
   are compressed to the front of their vector (permute by a 256-entry table) and stored at the running end of seq; the
   loop-carried chain is one popcount per 32 candidates.  A chunk with an order-dependent candidate (3 % of chunks at l = 20480,
   more lower down) and everything below i = 4096 takes the scalar steps. */
__attribute__((target("avx2,popcnt")))
static void accept_avx2(stream_t* s, int l, uint32_t* seq) {
    uint32_t i = (uint32_t)l - 1, mask = i;
    const uint32_t top = i;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    int pos = s->pos;
    while (i >= 1) {
        const uint32_t lo = mask >> 1;
        while (i > lo) {
            if (pos == MT_N) { stream_next_block(s); pos = 0; }
            const uint32_t* const w = s->w;
            if (lo >= 2047) {
                const __m256i vmask = _mm256_set1_epi32((int)mask);
                while (pos + 32 <= MT_N && i >= lo + 32) {
                    const __m256i sure = _mm256_set1_epi32((int)(i - 31));          /* v <  i - 31: accepted for certain */
                    const __m256i cur = _mm256_set1_epi32((int)i);                  /* v >  i     : rejected for certain */
                    __m256i v[4], acc[4];
                    __m256i decided = _mm256_set1_epi32(-1);
                    for (int t = 0; t < 4; ++t) {
                        v[t] = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(w + pos + 8 * t)), vmask);
                        acc[t] = _mm256_cmpgt_epi32(sure, v[t]);
                        decided = _mm256_and_si256(decided, _mm256_or_si256(acc[t], _mm256_cmpgt_epi32(v[t], cur)));
                    }
                    if (_mm256_movemask_ps(_mm256_castsi256_ps(decided)) != 0xff) {  /* order matters here: 32 scalar steps */
                        for (int k = 0; k < 32; ++k) {
                            const uint32_t x = w[pos + k] & mask;
                            seq[top - i] = x;
                            i = step_down(i, x);
                        }
                        pos += 32;
                        continue;
                    }
                    uint32_t n = top - i;
                    for (int t = 0; t < 4; ++t) {
                        const int m = _mm256_movemask_ps(_mm256_castsi256_ps(acc[t]));
                        const __m256i perm = _mm256_loadu_si256((const __m256i*)g_compress[m]);
                        _mm256_storeu_si256((__m256i*)(seq + n), _mm256_permutevar8x32_epi32(v[t], perm));
                        n += (uint32_t)__builtin_popcount((unsigned)m);
                    }
                    i = top - n;
                    pos += 32;
                }
            }
            while (pos + 8 <= MT_N && i >= lo + 8) {
                for (int k = 0; k < 8; ++k) {
                    const uint32_t x = w[pos + k] & mask;
                    seq[top - i] = x;
                    i = step_down(i, x);
                }
                pos += 8;
                if (lo >= 2047 && pos + 32 <= MT_N && i >= lo + 32) break;          /* back to the vector form */
            }
            while (pos < MT_N && i > lo && !(lo >= 2047 && pos + 32 <= MT_N && i >= lo + 32)) {
                const uint32_t x = w[pos++] & mask;
                seq[top - i] = x;
                i = step_down(i, x);
            }
        }
        mask = lo;
    }
    s->pos = pos;
}
#endif

/* One row: consumes the stream of np.random.choice(l, size=C, replace=False) and writes its C values.
   seq has room for l + 8 words (the vector stores run up to 7 lanes past the accepted ones). */
static void choice_row(stream_t* s, int l, int C, uint32_t* seq, int32_t* out, int vec) {
    if (l > 1) {
#if defined(__x86_64__)
        if (vec) accept_avx2(s, l, seq);
        else
#endif
            accept_scalar(s, l, seq);
    }
    /* seq[k] = j_i with i = l - 1 - k.  Walk the swaps backwards in time = i ascending = k descending. */
    if (C >= 1 && C <= WALK_MAX) {                                   /* the recipes' C = 2, 3: all tracks in one pass over seq */
        uint32_t q[WALK_MAX];
        for (int p = 0; p < C; ++p) q[p] = (uint32_t)p;
        int i = 1;
        for (; i < l && i <= C; ++i) {                               /* the head, where q == i can still happen */
            const uint32_t j = seq[l - 1 - i];
            for (int p = 0; p < C; ++p) {
                if (q[p] == (uint32_t)i) q[p] = j;
                else if (q[p] == j) q[p] = (uint32_t)i;
            }
        }
        walk_down(seq, l - 1 - i, l, C, q);                          /* q < i from here on: only j_i == q moves it, to i */
        for (int p = 0; p < C; ++p) out[p] = (int32_t)q[p];
        return;
    }
    for (int p = 0; p < C; ++p) {
        uint32_t q = (uint32_t)p;
        int i = 1;
        for (; i < l && i <= C; ++i) {
            const uint32_t j = seq[l - 1 - i];
            if (q == (uint32_t)i) q = j;
            else if (q == j) q = (uint32_t)i;
        }
        int k = l - 1 - i;
        while (k >= 0) {
            k = find_eq_down(seq, k, q);
            if (k >= 0) { q = (uint32_t)(l - 1 - k); --k; }
        }
        out[p] = (int32_t)q;
    }
}

/* key[624], *pos: numpy's MT19937 state, updated in place.  out: int32 [R, C].  Returns 0, or -1 on bad arguments
   (numpy raises for C > l: "Cannot take a larger sample than population when 'replace=False'"). */
int ams_mt_choice_rows(uint32_t* key, int32_t* pos, int R, int l, int C, int32_t* out) {
    if (!key || !pos || !out || R < 0 || l < 1 || C < 0 || C > l || *pos < 0 || *pos > MT_N) return -1;
    mt_t* g = (mt_t*)malloc(sizeof(mt_t));
    uint32_t* seq = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(l + 32));
    if (!g || !seq) { free(g); free(seq); return -2; }
    int vec = 0;
#if defined(__x86_64__)
    {
        const char* e = getenv("AMS_MT_CHOICE_SCALAR");              /* testing aid: every form is held against numpy */
        vec = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt") && !(e && e[0] == '1');
        if (vec && !g_compress_ready) compress_init();
    }
#endif
    memcpy(g->key, key, sizeof(g->key));
    g->pos = *pos;
    mt_temper(g);
    memset(seq, 0xff, sizeof(uint32_t) * (size_t)(l + 32));
    stream_t st = { g, g->out, g->pos };
    for (int r = 0; r < R; ++r) choice_row(&st, l, C, seq, out + (size_t)r * C, vec);
    memcpy(key, g->key, sizeof(g->key));
    *pos = st.pos;
    free(g); free(seq);
    return 0;
}
