/*
 * build_index seqfile indexfile -- build_index.c:15-107 on the MI355X library:
 * packs the sequence file exactly as the reference does (every byte of the
 * file is a base, newline included; exact-case C/G/T, anything else A; the
 * last len%4 bases read as A, build_index.c:80-87), builds the FM-index on
 * the device and writes the reference's index file byte for byte.
 */
#include <stdio.h>
#include <stdlib.h>

#include "bwt_mi355x.h"

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "Usage: %s seqfile indexfile\n", argv[0]);
    exit(1);
  }
  FILE *ifp = fopen(argv[1], "rb");
  if (!ifp) {
    fprintf(stderr, "Couldn't open index file\n"); /* sic, build_index.c:42 */
    exit(1);
  }
  FILE *ofp = fopen(argv[2], "w");
  if (!ofp) {
    fprintf(stderr, "Couldn't write to output file\n");
    exit(1);
  }
  fseek(ifp, 0L, SEEK_END);
  long len = ftell(ifp);
  rewind(ifp);
  char *ascii = malloc(len + 1);
  if (fread(ascii, 1, len, ifp) != (size_t)len) {
    fprintf(stderr, "Couldn't read sequence\n");
    exit(1);
  }
  fclose(ifp);
  char *seq = malloc(len / 4 + 1);
  fmi_pack_ascii(ascii, len, seq);
  free(ascii);
  printf("Finished reading sequence from file\n");
  fm_index *fmi = make_fmi_sacak(seq, (int)len); /* build_index.c:101 */
  if (!fmi) return 1;
  write_index(fmi, ofp);
  fclose(ofp);
  destroy_fmi(fmi);
  free(seq);
  return 0;
}
