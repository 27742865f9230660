/*
 * oracle/orc_sort.c -- CPU ORACLE (test infrastructure): klib-compatible UNSTABLE introsort.
 *
 * upstream bwa sorts seeds/chains/regions with ks_introsort; ties are resolved by the
 * mechanics of that algorithm, so the permutation (not just sortedness) is part of the
 * result.  This restates the behaviour of KSORT_INIT's ks_introsort/ks_combsort/
 * __ks_insertsort (spec: /root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/ksort.h:141-229)
 * for arbitrary element sizes: median-of-3 (first, middle+1, last) pivot moved to the end,
 * Hoare-style scan, partitions of <=16 left for one final insertion sort, comb sort when the
 * depth budget 2*ceil(log2 n) is exhausted.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct { char *l, *r; int depth; } frame_t;

#define EL(i) (base + (size_t)(i) * sz)
static inline void swp(char *a, char *b, size_t sz, char *t) { memcpy(t, a, sz); memcpy(a, b, sz); memcpy(b, t, sz); }

static void insertsort(char *s, char *t, size_t sz, orc_lt_f lt, char *tmp)
{	/* [s,t) */
	for (char *i = s + sz; i < t; i += sz)
		for (char *j = i; j > s && lt(j, j - sz); j -= sz) swp(j, j - sz, sz, tmp);
}

static void combsort(char *a, size_t n, size_t sz, orc_lt_f lt, char *tmp)
{
	const double shrink = 1.2473309501039786540366528676643;
	int do_swap; size_t gap = n;
	do {
		if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		do_swap = 0;
		for (char *i = a; i < a + (n - gap) * sz; i += sz) {
			char *j = i + gap * sz;
			if (lt(j, i)) { swp(i, j, sz, tmp); do_swap = 1; }
		}
	} while (do_swap || gap > 2);
	if (gap != 1) insertsort(a, a + n * sz, sz, lt, tmp);
}

void orc_introsort(void *base_, size_t n, size_t sz, orc_lt_f lt)
{
	char *a = base_, *s, *t, *i, *j, *k;
	char *rp = malloc(sz * 2), *tmp = rp + sz;
	int d; frame_t *stack, *top;
	if (n < 1) { free(rp); return; }
	if (n == 2) { if (lt(a + sz, a)) swp(a, a + sz, sz, tmp); free(rp); return; }
	for (d = 2; (1ul << d) < n; ++d);
	stack = malloc(sizeof(frame_t) * (sizeof(size_t) * d + 2));
	top = stack; s = a; t = a + (n - 1) * sz; d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { combsort(s, (t - s) / sz + 1, sz, lt, tmp); t = s; continue; }
			i = s; j = t; k = i + ((((j - i) / sz) >> 1) + 1) * sz;
			if (lt(k, i)) { if (lt(k, j)) k = j; }
			else k = lt(j, i) ? i : j;
			memcpy(rp, k, sz);
			if (k != t) swp(k, t, sz, tmp);
			for (;;) {
				do i += sz; while (lt(i, rp));
				do j -= sz; while (i <= j && lt(rp, j));
				if (j <= i) break;
				swp(i, j, sz, tmp);
			}
			swp(i, t, sz, tmp);
			if (i - s > t - i) {
				if ((size_t)(i - s) > 16 * sz) { top->l = s; top->r = i - sz; top->depth = d; ++top; }
				s = (size_t)(t - i) > 16 * sz ? i + sz : t;
			} else {
				if ((size_t)(t - i) > 16 * sz) { top->l = i + sz; top->r = t; top->depth = d; ++top; }
				t = (size_t)(i - s) > 16 * sz ? i - sz : s;
			}
		} else {
			if (top == stack) { free(stack); insertsort(a, a + n * sz, sz, lt, tmp); free(rp); return; }
			--top; s = top->l; t = top->r; d = top->depth;
		}
	}
}

static int lt_u64(const void *a, const void *b) { return *(const uint64_t*)a < *(const uint64_t*)b; }
void orc_introsort_u64(size_t n, uint64_t *a) { orc_introsort(a, n, 8, lt_u64); }
