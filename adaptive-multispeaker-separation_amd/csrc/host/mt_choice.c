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
     2. the rejection loop is branch-free: the candidate is stored at js[i] unconditionally and i only moves when it is accepted;
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

__attribute__((target_clones("avx2", "default")))
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

__attribute__((target_clones("avx2", "default")))
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

/* first index >= i0 with js[i] == q, or n */
__attribute__((target_clones("avx2", "default")))
static int find_eq(const uint32_t* js, int i0, int n, uint32_t q) {
    int i = i0;
    for (; i + 32 <= n; i += 32) {
        uint32_t any = 0;
        for (int k = 0; k < 32; ++k) any |= (js[i + k] == q);
        if (any) break;
    }
    for (; i < n; ++i)
        if (js[i] == q) return i;
    return n;
}

/* One row: consumes the stream of np.random.choice(l, size=C, replace=False) and writes its C values. */
static void choice_row(mt_t* s, int l, int C, uint32_t* js, int32_t* out) {
    if (l > 1) {
        uint32_t i = (uint32_t)l - 1, mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int pos = s->pos;
        while (i >= 1) {
            /* one mask level: i in (lo, mask].  Inside, the only loop-carried chain is compare + subtract on i; the candidate is
               stored unconditionally and i moves only when it is accepted (no data-dependent branch) */
            const uint32_t lo = mask >> 1;
            while (i > lo) {
                if (pos == MT_N) { mt_next_block(s); pos = 0; }
                const uint32_t* const w = s->out;
                /* chunks of 8 while neither bound can be crossed inside one (at most 8 acceptances, 8 words): the exit tests
                   leave the chain, which is step_down() */
                while (pos + 8 <= MT_N && i >= lo + 8) {
#pragma GCC unroll 8
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t v = w[pos + k] & mask;
                        js[i] = v;
                        i = step_down(i, v);
                    }
                    pos += 8;
                }
                while (pos < MT_N && i > lo) {
                    const uint32_t v = w[pos++] & mask;
                    js[i] = v;
                    i = step_down(i, v);
                }
            }
            mask = lo;                                               /* i == lo == 2^k - 1: the smallest mask >= i is lo itself */
        }
        s->pos = pos;
    }
    for (int p = 0; p < C; ++p) {
        uint32_t q = (uint32_t)p;
        int i = 1;
        for (; i < l && i <= C; ++i) {                               /* the head, where q == i can still happen */
            const uint32_t j = js[i];
            if (q == (uint32_t)i) q = j;
            else if (q == j) q = (uint32_t)i;
        }
        while (i < l) {                                              /* q < i from here on: only j_i == q moves it, to i */
            i = find_eq(js, i, l, q);
            if (i < l) { q = (uint32_t)i; ++i; }
        }
        out[p] = (int32_t)q;
    }
}

/* key[624], *pos: numpy's MT19937 state, updated in place.  out: int32 [R, C].  Returns 0, or -1 on bad arguments
   (numpy raises for C > l: "Cannot take a larger sample than population when 'replace=False'"). */
int ams_mt_choice_rows(uint32_t* key, int32_t* pos, int R, int l, int C, int32_t* out) {
    if (!key || !pos || !out || R < 0 || l < 1 || C < 0 || C > l || *pos < 0 || *pos > MT_N) return -1;
    mt_t* s = (mt_t*)malloc(sizeof(mt_t));
    uint32_t* js = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(l + 32));
    if (!s || !js) { free(s); free(js); return -2; }
    memcpy(s->key, key, sizeof(s->key));
    s->pos = *pos;
    mt_temper(s);
    memset(js, 0xff, sizeof(uint32_t) * (size_t)(l + 32));
    for (int r = 0; r < R; ++r) choice_row(s, l, C, js, out + (size_t)r * C);
    memcpy(key, s->key, sizeof(s->key));
    *pos = s->pos;
    free(s); free(js);
    return 0;
}
