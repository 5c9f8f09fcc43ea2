/*
 * gen_seq [x [seed]] -- gen_seq.c: x (default 100000) characters drawn as
 * "ACGT"[rand() & 3], no trailing newline.  The reference seeds with time(0)
 * (gen_seq.c:9); an explicit seed makes a run reproducible.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

int main(int argc, char **argv) {
  long x = argc > 1 ? atol(argv[1]) : 100000;
  srand(argc > 2 ? (unsigned)atol(argv[2]) : (unsigned)time(0));
  for (long i = 0; i < x; ++i) putchar("ACGT"[rand() & 3]);
  return 0;
}
