/* sort_tmpl.h -- bsx_introsort (util.c) for one element type and one order, compiled in place: the same sequence of comparisons and
 * swaps (klib's ks_introsort: median-of-3 quicksort that leaves runs of <= 16 for one final insertion sort, comb sort when the depth
 * budget is spent), so the same result for equal keys too, without a call per comparison and three memcpy()s per swap.  The sorts of
 * the back half (mem_sort_dedup_patch's two, mem_mark_primary_se's, mem_pair's) were a third of the host's CPU time on a repeat-rich
 * genome (round 5, tools/dbg/host_prof.sh).
 *   BSX_SORT_DEFINE(name, type, LT)      LT(a, b): an expression over two `const type *`  ->  static void name(size_t n, type *a) */
#ifndef BSX_SORT_TMPL_H
#define BSX_SORT_TMPL_H
#include <stddef.h>

#define BSX_SORT_DEFINE(NAME, T, LT) \
static inline void NAME##_ins(T *a, ptrdiff_t s, ptrdiff_t t) /* [s, t): swaps of neighbours, written as one shift */ \
{ \
	ptrdiff_t i, j; \
	for (i = s + 1; i < t; ++i) { \
		if (LT(&a[i], &a[i - 1])) { \
			T v = a[i]; \
			for (j = i; j > s && LT(&v, &a[j - 1]); --j) a[j] = a[j - 1]; \
			a[j] = v; \
		} \
	} \
} \
static void NAME##_comb(T *a, ptrdiff_t s, size_t n) /* ksort.h:162-183 */ \
{ \
	const double shrink_factor = 1.2473309501039786540366528676643; \
	int swapped; \
	size_t gap = n, i; \
	do { \
		if (gap > 2) { \
			gap = (size_t)(gap / shrink_factor); \
			if (gap == 9 || gap == 10) gap = 11; \
		} \
		swapped = 0; \
		for (i = 0; i + gap < n; ++i) \
			if (LT(&a[s + i + gap], &a[s + i])) { T v = a[s + i]; a[s + i] = a[s + i + gap]; a[s + i + gap] = v; swapped = 1; } \
	} while (swapped || gap > 2); \
	if (gap != 1) NAME##_ins(a, s, s + (ptrdiff_t)n); \
} \
static void NAME(size_t n, T *a) \
{ \
	struct { ptrdiff_t left, right; int depth; } stack[sizeof(size_t) * 64 + 2], *top = stack; \
	ptrdiff_t s, t, i, j, k; \
	int d; \
	T pivot, sw; \
	if (n < 2) return; \
	if (n == 2) { if (LT(&a[1], &a[0])) { sw = a[0]; a[0] = a[1]; a[1] = sw; } return; } \
	for (d = 2; 1ul << d < n; ++d); \
	s = 0; t = (ptrdiff_t)n - 1; d <<= 1; \
	for (;;) { \
		if (s < t) { \
			if (--d == 0) { NAME##_comb(a, s, (size_t)(t - s + 1)); t = s; continue; } \
			i = s; j = t; k = i + ((j - i) >> 1) + 1; \
			if (LT(&a[k], &a[i])) { if (LT(&a[k], &a[j])) k = j; } \
			else k = LT(&a[j], &a[i]) ? i : j; \
			pivot = a[k]; \
			if (k != t) { sw = a[k]; a[k] = a[t]; a[t] = sw; } \
			for (;;) { \
				do ++i; while (LT(&a[i], &pivot)); \
				do --j; while (i <= j && LT(&pivot, &a[j])); \
				if (j <= i) break; \
				sw = a[i]; a[i] = a[j]; a[j] = sw; \
			} \
			sw = a[i]; a[i] = a[t]; a[t] = sw; \
			if (i - s > t - i) { \
				if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; } \
				s = t - i > 16 ? i + 1 : t; \
			} else { \
				if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; } \
				t = i - s > 16 ? i - 1 : s; \
			} \
		} else { \
			if (top == stack) { NAME##_ins(a, 0, (ptrdiff_t)n); return; } \
			--top; s = top->left; t = top->right; d = top->depth; \
		} \
	} \
}
#endif
