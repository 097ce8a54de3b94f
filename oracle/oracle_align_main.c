/* oracle/oracle_align_main.c -- TEST INFRASTRUCTURE: `biscuit align` running the host pipeline over the
 * CPU restatement of the kernels (oracle/port.c).  Used as the SAM-level checker for the HIP path
 * and as bench.py's timed cpu_baseline ("port").  Never part of the product. */
int oracle_align_main(int argc, char **argv);
int main(int argc, char **argv) { return oracle_align_main(argc, argv); }
