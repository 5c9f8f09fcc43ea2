/*
 * search_reads seqfile indexfile readfile -- search_reads.c:25-170 on the
 * MI355X library.  Same command line and the same report lines.  Two things
 * differ from the program as shipped, both deliberate:
 *   - reads are encoded before the search as single_align.c:450-474 encodes
 *     them (A C G T -> 0 1 2 3, anything else N); the shipped program hands raw
 *     ASCII to mms, which indexes C[] with it and crashes on ACGT input;
 *   - all reads of the file go to the device in one batch, the per-read loop
 *     of search_reads.c:115-153 runs one read per lane.
 * The sequence file is not needed to search (read_index ignores it, as in the
 * reference); it is packed as build_index packs it and handed to the device,
 * which uses it to speed the searches up when it is the text of the index.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bwt_mi355x.h"

int main(int argc, char **argv) {
  if (argc != 4) {
    fprintf(stderr, "Usage: %s seqfile indexfile readfile\n", argv[0]);
    exit(-1);
  }
  FILE *sfp = fopen(argv[1], "rb");
  if (!sfp) {
    fprintf(stderr, "Could not open sequence\n");
    exit(-1);
  }
  fseek(sfp, 0L, SEEK_END);
  long slen = ftell(sfp);
  rewind(sfp);
  char *ascii = malloc(slen + 1), *seq = malloc(slen / 4 + 1);
  if (ascii && seq && fread(ascii, 1, slen, sfp) == (size_t)slen) {
    fmi_pack_ascii(ascii, slen, seq);
  } else {
    free(seq);
    seq = NULL;
  }
  free(ascii);
  fclose(sfp);
  FILE *ifp = fopen(argv[2], "rb");
  if (!ifp) {
    fprintf(stderr, "Could not open index file");
    exit(-1);
  }
  fm_index *fmi = read_index(seq, ifp);
  fclose(ifp);
  if (!fmi) return 1;
  if (seq && fmi->len == (int)slen) fmi_attach_text(fmi, seq); /* a text that does not fit is left out */
  free(seq);
  FILE *rfp = fopen(argv[3], "r");
  if (!rfp) {
    fprintf(stderr, "Could not open reads file");
    exit(-1);
  }
  /* every line is a read (search_reads.c:102-111) */
  size_t cap = 1 << 20, used = 0, nreads = 0, ocap = 1 << 16;
  char *text = malloc(cap);
  unsigned long *offsets = malloc(ocap * sizeof *offsets);
  char *line = malloc(256 * 256);
  offsets[0] = 0;
  while (fgets(line, 256 * 256 - 1, rfp)) {
    size_t len = strlen(line);
    if (len && line[len - 1] == '\n') line[--len] = 0;
    if (used + len + 1 > cap) {
      while (used + len + 1 > cap) cap *= 2;
      text = realloc(text, cap);
    }
    if (nreads + 2 > ocap) {
      ocap *= 2;
      offsets = realloc(offsets, ocap * sizeof *offsets);
    }
    memcpy(text + used, line, len);
    used += len;
    offsets[++nreads] = used;
  }
  fclose(rfp);
  char *enc = malloc(used + 1);
  fmi_encode_read(text, (long)used, enc);
  int *out = malloc((nreads ? nreads : 1) * 4 * sizeof *out);
  printf("Beginning alignment\n");
  if (fmi_search_reads_batch(fmi, enc, offsets, (long)nreads, 20, out)) return 1;
  for (size_t r = 0; r < nreads; ++r) {
    int fwd = out[4 * r], bwd = out[4 * r + 1], fpos = out[4 * r + 2], bpos = out[4 * r + 3];
    if (fwd && bwd && abs(fpos - bpos) < 10000) { /* search_reads.c:154-159 */
      printf("\nRead %zu: Aligned both forward (%d) and backward (%d)\n", r, fwd, bwd);
      printf("At locations %d and %d respectively\n", fpos, bpos);
      printf("%.*s\n", (int)(offsets[r + 1] - offsets[r]), text + offsets[r]);
    }
  }
  destroy_fmi(fmi);
  free(text);
  free(enc);
  free(offsets);
  free(line);
  free(out);
  return 0;
}
