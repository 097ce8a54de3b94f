/* prof.c -- $BSX_PROF_SAMPLE=<file>: a sampling profile of the host side of a run (there is no perf on the GPU boxes).  A process-CPU-time
 * timer (ITIMER_PROF, 1 kHz) interrupts whichever thread is running; the handler notes the interrupted instruction's address and the
 * and at exit the samples are written as "count offset object" lines (tools/prof_symbols.py names the functions).  Debug facility: off
 * unless the variable is set. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
#include <ucontext.h>
#include "bsx_core.h"

#define PROF_MAX (1 << 22)
static void **g_pc, **g_ra; static volatile long g_n; static const char *g_path;
static char *g_lo, *g_hi;   /* this library, roughly: 64 MB from its base is enough to tell it from libc */

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
	ucontext_t *uc = (ucontext_t*)uc_;
	long k = __sync_fetch_and_add(&g_n, 1);
	(void)sig; (void)si;
	if (k < PROF_MAX) {
		g_pc[k] = (void*)uc->uc_mcontext.gregs[REG_RIP];
		/* a sample outside this library (memcpy, memset ...: leaf routines that push nothing): the word on top of the stack is the return
		 * address into whoever called it -- a guess, kept only if it points back into this library */
		g_ra[k] = 0;
		if ((char*)g_pc[k] < g_lo || (char*)g_pc[k] >= g_hi) { /* the nearest word up the stack that points into this library (stale words can mislead: a guess) */
			void **sp = (void**)uc->uc_mcontext.gregs[REG_RSP];
			int w;
			for (w = 0; w < 96; ++w) if ((char*)sp[w] >= g_lo && (char*)sp[w] < g_hi) { g_ra[k] = sp[w]; break; }
		}
	}
}
static int cmp_ptr(const void *a, const void *b) { const void *x = *(void* const*)a, *y = *(void* const*)b; return x < y ? -1 : x > y; }
/* "count offset object" per distinct address (offset inside its shared object: tools/prof_symbols.py turns them into functions with nm) */
static void prof_dump(void)
{
	struct itimerval off; long n = g_n < PROF_MAX ? g_n : PROF_MAX, i;
	FILE *f;
	memset(&off, 0, sizeof(off));
	setitimer(ITIMER_PROF, &off, 0);
	if (!g_path || n == 0) return;
	for (i = 0; i < n; ++i) if (g_ra[i] && (char*)g_ra[i] >= g_lo && (char*)g_ra[i] < g_hi) g_pc[i] = (void*)((unsigned long)g_ra[i] | 1ul << 62);   /* marked: "called from" */
	qsort(g_pc, (size_t)n, sizeof(void*), cmp_ptr);
	f = fopen(g_path, "w");
	if (!f) return;
	fprintf(f, "# %ld samples at 1 kHz of process CPU time\n", n);
	for (i = 0; i < n; ) {
		long j = i; Dl_info di; const char *fl = "?"; unsigned long base = 0;
		while (j < n && g_pc[j] == g_pc[i]) ++j;
		const int from = (int)((unsigned long)g_pc[i] >> 62 & 1);
		void *pc = (void*)((unsigned long)g_pc[i] & ~(1ul << 62));
		if (dladdr(pc, &di) && di.dli_fname) { fl = di.dli_fname; base = (unsigned long)di.dli_fbase; }
		fprintf(f, "%ld %lx %s%s\n", j - i, (unsigned long)pc - base, from ? "libc<-" : "", fl);
		i = j;
	}
	fclose(f);
}
__attribute__((constructor)) static void prof_init(void)
{
	struct sigaction sa; struct itimerval it;
	g_path = getenv("BSX_PROF_SAMPLE");
	if (!g_path || !*g_path) { g_path = 0; return; }
	if (strstr(g_path, "%d")) { /* a file per process (a launcher and the command line it starts both load this library) */
		static char per_pid[1024];
		snprintf(per_pid, sizeof(per_pid), g_path, (int)getpid());
		g_path = per_pid;
	}
	g_pc = (void**)malloc(sizeof(void*) * PROF_MAX); g_ra = (void**)malloc(sizeof(void*) * PROF_MAX);
	{ Dl_info di; if (dladdr((void*)prof_dump, &di) && di.dli_fbase) { g_lo = (char*)di.dli_fbase; g_hi = g_lo + (64l << 20); } }
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, 0);
	it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000; it.it_value = it.it_interval;
	setitimer(ITIMER_PROF, &it, 0);
	atexit(prof_dump);
}
