/* tune.h -- the library's tuning and test knobs, one registry (tune.c).
 *
 * Until round 5 each of these was an environment variable of its own, read where it was used -- three dozen names, some of them looked up
 * with getenv() on every kernel launch.  They are named settings of the library now: set through bsx_tune_set() (tests, bench.py, the A/B
 * tools) or, for a whole process, through the ONE environment variable $BSX_TUNE ("name=value,name=value", parsed once).  None of them
 * changes the output; defaults are what the product runs with.  The list of names, with what each does, is the table in tune.c. */
#ifndef BSX_TUNE_H
#define BSX_TUNE_H
#ifdef __cplusplus
extern "C" {
#endif
/* value == NULL: back to the default.  BSX_E_ARG for a name that is not in the table. */
int bsx_tune_set(const char *name, const char *value);
/* the setting as text; NULL when it is not set (the caller's default applies) */
const char *bsx_tune_str(const char *name);
long bsx_tune_long(const char *name, long dflt);
int bsx_tune_is_set(const char *name);
/* $BSX_PHASES / "phases": 0 off, 1 per-phase lines and kernel cycle counters, 2 per-tier counter read-outs */
int bsx_phases(void);
/* names and one-line descriptions, for the documentation: entry i, or NULL past the end */
const char *bsx_tune_name(int i);
const char *bsx_tune_doc(int i);
#ifdef __cplusplus
}
#endif
#endif
