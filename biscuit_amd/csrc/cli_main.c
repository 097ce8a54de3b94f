/* cli_main.c -- `biscuit_align`: stands in for `biscuit align` (src/main.c:105-159 builds the @PG line
 * the same way: VN + the full command line). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bsx.h"
extern char *bsx_pg_line;
int main(int argc, char **argv)
{
	size_t l = 64; int i;
	char *pg;
	for (i = 0; i < argc; ++i) l += strlen(argv[i]) + 1;
	pg = (char*)malloc(l + strlen(bsx_version()));
	sprintf(pg, "@PG\tID:biscuit\tPN:biscuit\tVN:%s\tCL:%s", bsx_version(), argv[0]);
	for (i = 1; i < argc; ++i) { strcat(pg, " "); strcat(pg, argv[i]); }
	bsx_pg_line = pg;
	/* accept both `biscuit_align align ...` and `biscuit_align ...` */
	if (argc > 1 && strcmp(argv[1], "align") == 0) return bsx_align_main(argc - 1, argv + 1);
	return bsx_align_main(argc, argv);
}
