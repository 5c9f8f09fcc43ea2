/*
 * single_align seqfile indexfile readfile -- the main() of single_align.c
 * (:364-598) on the MI355X library: same command line, same output (for every
 * read the position + 1 and the CIGAR as stack_print_destroy prints it, or 0;
 * on stderr the count of aligned reads).  What differs: all reads of the file
 * are aligned in one batch (one wave per read on the device), and the reads
 * that do not align forward go through a second batch as reverse complements
 * (single_align.c:489-503 does the same one read at a time).  Where the
 * reference's align_read_anchored is undefined (it calls mms on no bases) the
 * read counts as not aligned in that direction.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bwt_mi355x.h"

#define ANCHOR_LEN 12 /* single_align.c:483 */

/* stack_print_destroy (stack.c:27-37) of the stack the unit operations stand for:
 * a space, then the runs from the last pushed to the first */
static void print_cigar(const char *ops, unsigned n) {
  putchar(' ');
  unsigned end = n;
  while (end) {
    unsigned begin = end - 1;
    while (begin && ops[begin - 1] == ops[end - 1]) --begin;
    printf("%u%c", end - begin, ops[end - 1]);
    end = begin;
  }
  printf("\n");
}

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
  if (!ascii || !seq || fread(ascii, 1, slen, sfp) != (size_t)slen) {
    fprintf(stderr, "Could not read sequence\n");
    exit(-1);
  }
  fmi_pack_ascii(ascii, slen, seq); /* single_align.c:385-419 packs as build_index.c:54-87 does */
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
  if (fmi->len == (int)slen) fmi_attach_text(fmi, seq); /* faster searches; a text that does not fit is left out */
  FILE *rfp = fopen(argv[3], "r");
  if (!rfp) {
    fprintf(stderr, "Could not open reads file");
    exit(-1);
  }
  /* every line is a read (single_align.c:441-449) */
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
  free(line);
  const size_t nr = nreads ? nreads : 1;
  char *fwd = malloc(used + 1);
  fmi_encode_read(text, (long)used, fwd); /* single_align.c:450-474 */
  int *pos = malloc(nr * sizeof *pos);
  unsigned *nops = malloc(nr * sizeof *nops);
  char *ops = malloc(5 * used + 64 * nr);
  if (nreads && fmi_align_reads_batch(fmi, seq, fwd, offsets, (long)nreads, ANCHOR_LEN, pos, ops, nops)) return 1;
  /* the reads that did not align: their reverse complements (single_align.c:455-471, :493) */
  size_t nrev = 0, rev_used = 0;
  for (size_t r = 0; r < nreads; ++r)
    if (pos[r] == 0 || pos[r] == INT_MIN) {
      ++nrev;
      rev_used += offsets[r + 1] - offsets[r];
    }
  char *rev = malloc(rev_used + 1), *rops = malloc(5 * rev_used + 64 * (nrev ? nrev : 1));
  unsigned long *roff = malloc((nrev + 1) * sizeof *roff);
  int *rpos = malloc((nrev ? nrev : 1) * sizeof *rpos);
  unsigned *rnops = malloc((nrev ? nrev : 1) * sizeof *rnops);
  size_t k = 0;
  roff[0] = 0;
  for (size_t r = 0; r < nreads; ++r)
    if (pos[r] == 0 || pos[r] == INT_MIN) {
      const unsigned long a = offsets[r], len = offsets[r + 1] - a;
      for (unsigned long i = 0; i < len; ++i) {
        const char c = fwd[a + i];
        rev[roff[k] + len - 1 - i] = c == 5 ? 5 : (char)(3 - c);
      }
      roff[k + 1] = roff[k] + len;
      ++k;
    }
  if (nrev && fmi_align_reads_batch(fmi, seq, rev, roff, (long)nrev, ANCHOR_LEN, rpos, rops, rnops)) return 1;
  int naligned = 0;
  k = 0;
  for (size_t r = 0; r < nreads; ++r) {
    if (pos[r] != 0 && pos[r] != INT_MIN) {
      naligned++;
      printf("%d\n", pos[r] + 1);
      print_cigar(ops + 5 * offsets[r] + 64 * r, nops[r]);
      continue;
    }
    if (rpos[k] != 0 && rpos[k] != INT_MIN) {
      naligned++;
      printf("%d\n", rpos[k] + 1);
      print_cigar(rops + 5 * roff[k] + 64 * k, rnops[k]);
    } else {
      printf("0\n");
    }
    ++k;
  }
  fprintf(stderr, "%d of %d reads aligned\n", naligned, (int)nreads);
  destroy_fmi(fmi);
  free(seq);
  free(text);
  free(fwd);
  free(rev);
  free(offsets);
  free(roff);
  free(pos);
  free(rpos);
  free(nops);
  free(rnops);
  free(ops);
  free(rops);
  return 0;
}
