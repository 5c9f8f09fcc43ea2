/*
 * fmitest len -- the reference's build + search benchmark (fmitest.c:100-173)
 * on the MI355X library: same command line, same random inputs (glibc rand,
 * unseeded, as fmitest.c:127-153), same two report lines.  The index comes
 * from make_fmi; the 10,000,000 overlapping 12-mers that the reference hands
 * to four pthreads (fmitest.c:54-91) go to the device in one batch.
 * Times are wall clock; the "cycles" figure is the TSC delta, as there.
 * A last line says how many of the 12-mers occur in the text; with a second
 * argument the text and the patterns are written to that file (len/4+1 bytes,
 * then 10,000,011), so that a test can count for itself: the rand() stream is
 * not the same from process to process once a GPU runtime has been loaded.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <x86intrin.h>

#include "bwt_mi355x.h"

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv) {
  if (argc != 2 && argc != 3) {
    fprintf(stderr, "Usage: fmitest len\n");
    exit(-1);
  }
  int len = atoi(argv[1]);
  if (len <= 0) {
    fprintf(stderr, "Usage: fmitest len\n");
    exit(-1);
  }
  char *str = malloc(len / 4 + 1);
  for (int i = 0; i < len / 4 + 1; ++i) str[i] = (char)rand();
  /* base len must read 0 (fmitest.c:130-142) */
  static const unsigned char keep[4] = {0x00, 0xC0, 0xF0, 0xFC};
  str[len / 4] &= (char)keep[len % 4];

  unsigned long long a = __rdtsc();
  double t0 = now();
  fm_index *fmi = make_fmi(str, len);
  double t1 = now();
  unsigned long long b = __rdtsc();
  if (!fmi) return 1;
  printf("Built index with %d base pairs in %lld cycles (%f s)\n", len, (long long)(b - a), t1 - t0);
  printf("(%f cycles per base pair (%e seconds))\n", (double)(b - a) / len, (t1 - t0) / len);

  const long npat = 10000000;
  char *pats = malloc(npat + 11);
  for (long i = 0; i < npat + 11; ++i) pats[i] = rand() & 3;
  int *res = malloc(npat * sizeof *res);
  a = __rdtsc();
  t0 = now();
  int rc = fmi_reverse_search_batch(fmi, pats, 12, 1, npat, res);
  t1 = now();
  b = __rdtsc();
  if (rc) return 1;
  printf("Searched 10000000 12bp sequences in %lld cycles (%f s)\n", (long long)(b - a), t1 - t0);
  printf("(%f cycles per base pair (%e seconds))\n", (double)(b - a) / 120000000., (t1 - t0) / 120000000.);
  long hits = 0;
  for (long i = 0; i < npat; ++i) hits += res[i] > 1;
  printf("(%ld of them occur in the text)\n", hits);
  if (argc == 3) {
    FILE *f = fopen(argv[2], "wb");
    if (!f || fwrite(str, 1, len / 4 + 1, f) != (size_t)(len / 4 + 1) || fwrite(pats, 1, npat + 11, f) != (size_t)(npat + 11) ||
        fclose(f)) {
      fprintf(stderr, "Could not write %s\n", argv[2]);
      return 1;
    }
  }

  destroy_fmi(fmi);
  free(str);
  free(pats);
  free(res);
  return 0;
}
