/*
 * The reference's C API (histsortcomp.h, csacak.h, seqindex.h, fileio.h) over
 * the HIP kernels of libbwt_mi355x.so.
 *
 * Everything that is a batch of work -- building a suffix array, a BWT, an
 * index, loading one, and every fmi_*_batch call -- is handed to a device entry
 * point of bwt_gpu.h; there is no host implementation of those to fall back on,
 * and when the device call fails the error goes to stderr and the function
 * returns the reference's failure value.
 *
 * The reference's single-query functions (rank, lf, unc_sa, loc_search,
 * reverse_search, locate, mms) answer ONE query per call: a few hundred
 * nanoseconds of work on arrays the fm_index already holds on the host (bwt,
 * idxs, rank_index), against 5-17 microseconds for a kernel launch and a
 * synchronisation.  They are computed on the host from those arrays (the part
 * of this file headed "single queries"), so that a caller that loops over them
 * -- the reference's own fmitest.c:54-65, searchtest.c:71-89 -- is not slowed
 * down by the library.  BWT_MI355X_SCALAR=device sends them to the GPU as
 * batches of one instead (the tests compare the two), and so does an fm_index
 * that lacks the host rank table (len >= 2^28).  Callers that want throughput
 * use the batch forms.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bwt_gpu.h"
#include "bwt_mi355x.h"

/* ------------------------------------------------------------------ */
/* device selection and the fm_index -> device index registry          */
/* ------------------------------------------------------------------ */

/* The GPUs this layer uses.  BWT_MI355X_DEVICES = "all" or a comma-separated list
 * of ordinals (an ordinal may be listed more than once: each mention is a slot
 * with a replica of its own, which is how the partitioning is tested on a
 * one-GPU machine); else BWT_MI355X_DEVICE = one ordinal; else every visible GPU.
 * Slot 0 is the primary: builds, loads, single queries and small batches run
 * there.  The other slots get a replica of an index the first time a batch is
 * large enough to be split (fmi_*_batch below), and from then on take their
 * contiguous share of every such batch: the static split over worker threads of
 * fmitest.c:67-91, with a GPU behind each worker.  A process that is itself one
 * of several (one per GPU) sets BWT_MI355X_DEVICE to its own GPU. */
#define FMI_MAX_SLOTS 16
static int g_slot_dev[FMI_MAX_SLOTS];
static int g_nslots;
static long g_min_reads_per_slot = 65536;
static pthread_once_t g_slots_once = PTHREAD_ONCE_INIT;

static void slots_init(void) {
  const char *list = getenv("BWT_MI355X_DEVICES");
  const char *one = getenv("BWT_MI355X_DEVICE");
  const char *mr = getenv("BWT_MI355X_MIN_READS_PER_DEVICE");
  if (mr && atol(mr) > 0) g_min_reads_per_slot = atol(mr);
  if (list && strcmp(list, "all") != 0) {
    const char *p = list;
    while (*p && g_nslots < FMI_MAX_SLOTS) {
      char *end;
      long d = strtol(p, &end, 10);
      if (end == p) break;
      g_slot_dev[g_nslots++] = (int)d;
      p = *end == ',' ? end + 1 : end;
    }
  } else if (!list && one) {
    g_slot_dev[g_nslots++] = atoi(one);
  } else {
    int n = bwtgpu_device_count();
    for (int d = 0; d < n && g_nslots < FMI_MAX_SLOTS; ++d) g_slot_dev[g_nslots++] = d;
  }
  if (g_nslots == 0) g_slot_dev[g_nslots++] = 0; /* bwtgpu_init then says what is wrong */
}

static int slot_ready(int slot) {
  /* every host thread that enters the API makes its device current */
  pthread_once(&g_slots_once, slots_init);
  int rc = bwtgpu_init(g_slot_dev[slot]);
  if (rc) fprintf(stderr, "bwt_mi355x: %s\n", bwtgpu_last_error());
  return rc;
}

static int device_ready(void) { return slot_ready(0); }

int fmi_device_slots(void) {
  pthread_once(&g_slots_once, slots_init);
  return g_nslots;
}

static void report(const char *what) {
  fprintf(stderr, "bwt_mi355x: %s: %s\n", what, bwtgpu_last_error());
}

struct reg_entry {
  const fm_index *fmi;
  bwtgpu_index *ix;                      /* on slot 0 */
  bwtgpu_index *replica[FMI_MAX_SLOTS];  /* [0] unused; made on first use of the slot */
  pthread_mutex_t replica_lock[FMI_MAX_SLOTS];
  int *rank_slab; /* rows of rank_index, one allocation */
  struct reg_entry *next;
};
static struct reg_entry *g_reg;
static pthread_mutex_t g_reg_lock = PTHREAD_MUTEX_INITIALIZER;

static void reg_put(const fm_index *fmi, bwtgpu_index *ix, int *rank_slab) {
  struct reg_entry *e = calloc(1, sizeof *e);
  e->fmi = fmi;
  e->ix = ix;
  e->rank_slab = rank_slab;
  for (int g = 0; g < FMI_MAX_SLOTS; ++g) pthread_mutex_init(&e->replica_lock[g], NULL);
  pthread_mutex_lock(&g_reg_lock);
  e->next = g_reg;
  g_reg = e;
  pthread_mutex_unlock(&g_reg_lock);
}

static struct reg_entry *reg_find_locked(const fm_index *fmi) {
  for (struct reg_entry *e = g_reg; e; e = e->next)
    if (e->fmi == fmi) return e;
  return NULL;
}

/* The device index behind an fm_index.  One that did not come from make_fmi
 * or read_index (a caller filled the struct itself) is uploaded on first use
 * from its stored parts, as read_index would. */
static struct reg_entry *device_entry(const fm_index *fmi) {
  if (!fmi) return NULL;
  pthread_mutex_lock(&g_reg_lock);
  struct reg_entry *e = reg_find_locked(fmi);
  pthread_mutex_unlock(&g_reg_lock);
  if (device_ready()) return NULL;
  if (e) return e;
  bwtgpu_index *ix = bwtgpu_index_from_parts((const uint8_t *)fmi->bwt, (uint32_t)fmi->len, (uint32_t)fmi->endloc,
                                             (const uint32_t *)fmi->idxs, NULL);
  if (!ix) {
    report("index upload");
    return NULL;
  }
  reg_put(fmi, ix, NULL);
  pthread_mutex_lock(&g_reg_lock);
  e = reg_find_locked(fmi);
  pthread_mutex_unlock(&g_reg_lock);
  return e;
}

static bwtgpu_index *device_index(const fm_index *fmi) {
  struct reg_entry *e = device_entry(fmi);
  return e ? e->ix : NULL;
}

/* the index of slot g > 0: a copy of the primary in that GPU's memory, made once */
static bwtgpu_index *replica_index(struct reg_entry *e, int g) {
  pthread_mutex_lock(&e->replica_lock[g]);
  if (!e->replica[g]) {
    e->replica[g] = bwtgpu_index_replicate(e->ix, g_slot_dev[g]);
    if (!e->replica[g]) report("index replica");
  }
  bwtgpu_index *ix = e->replica[g];
  pthread_mutex_unlock(&e->replica_lock[g]);
  return ix;
}

/* A batch of `count` independent items over the slots: slot g takes the contiguous
 * share fmi_share_bounds gives it, each share on a host thread of its own with
 * its GPU current, the caller's thread taking share 0.  `fn` runs one share on one
 * index and returns 0 or -1; results land in the caller's arrays at the share's
 * offset, so they come out in item order.  Batches too small to give every slot
 * min_reads_per_slot items use fewer slots, down to the primary alone. */
typedef int (*share_fn)(bwtgpu_index *ix, long lo, long hi, void *ctx);
struct share_job {
  struct reg_entry *e;
  int slot;
  long lo, hi;
  share_fn fn;
  void *ctx;
  int rc;
  pthread_t th;
};

void fmi_share_bounds(long count, int shares, int g, long *lo, long *hi) {
  const long base = count / shares, extra = count % shares; /* the first `extra` shares hold one item more */
  *lo = g * base + (g < extra ? g : extra);
  *hi = *lo + base + (g < extra ? 1 : 0);
}

static void *share_worker(void *arg) {
  struct share_job *j = arg;
  j->rc = -1;
  if (slot_ready(j->slot)) return NULL;
  bwtgpu_index *ix = replica_index(j->e, j->slot);
  if (ix) j->rc = j->fn(ix, j->lo, j->hi, j->ctx);
  return NULL;
}

static int run_shared(const fm_index *fmi, long count, share_fn fn, void *ctx) {
  struct reg_entry *e = device_entry(fmi);
  if (!e) return -1;
  int G = g_nslots;
  if (count < g_min_reads_per_slot * (long)G) G = (int)(count / g_min_reads_per_slot);
  if (G <= 1) return fn(e->ix, 0, count, ctx);
  struct share_job job[FMI_MAX_SLOTS];
  for (int g = 0; g < G; ++g) {
    job[g] = (struct share_job){.e = e, .slot = g, .fn = fn, .ctx = ctx, .rc = -1};
    fmi_share_bounds(count, G, g, &job[g].lo, &job[g].hi);
  }
  int started[FMI_MAX_SLOTS] = {0};
  for (int g = 1; g < G; ++g) started[g] = pthread_create(&job[g].th, NULL, share_worker, &job[g]) == 0;
  int rc = fn(e->ix, job[0].lo, job[0].hi, ctx);
  for (int g = 1; g < G; ++g) {
    if (started[g]) pthread_join(job[g].th, NULL);
    else share_worker(&job[g]), device_ready(); /* no thread to be had: the share runs here, then back to the primary */
    if (job[g].rc) rc = -1;
  }
  return rc;
}

/* ------------------------------------------------------------------ */
/* suffix array and BWT (histsortcomp.h, csacak.h)                     */
/* ------------------------------------------------------------------ */

static uint32_t *sa_on_device(const char *str, uint32_t len) {
  if (device_ready()) return NULL;
  uint32_t *sa = malloc(((size_t)len + 1) * sizeof *sa);
  if (!sa) return NULL;
  if (bwtgpu_sa_build_host((const uint8_t *)str, len, sa)) {
    report("suffix array");
    free(sa);
    return NULL;
  }
  return sa;
}

int *histsort(const char *str, int len) { return (int *)sa_on_device(str, (uint32_t)len); }

unsigned int *csuff_arr(const char *seq, unsigned int len) { return sa_on_device(seq, len); }

void histhelper(const char *base, int len, int *arr, int *aux, int start, int end, int depth) {
  /* histsortcomp.c:35-90: arr[start .. end) are starts of suffixes that agree on
   * their first `depth` bases; sorts them by what follows.  The reference leaves
   * the result in whichever of the two arrays its file-static `saux` is not
   * (histsortcomp.c:25, :45-50), which only histsort itself can know; here both
   * hold it.  `len` is the number of bases, as histsort passes it (:171). */
  (void)depth;
  if (end - start <= 0) return;
  if (end - start > 1) {
    if (device_ready()) return;
    if (bwtgpu_sort_suffixes_host((const uint8_t *)base, (uint32_t)len, (uint32_t *)arr + start,
                                  (uint64_t)(end - start))) {
      report("histhelper");
      return;
    }
  }
  if (aux) memcpy(aux + start, arr + start, (size_t)(end - start) * sizeof *arr);
}

unsigned int sprintcbwt(const char *str, unsigned int *idxs, unsigned int len, char *out) {
  uint32_t endloc = 0xFFFFFFFFu;
  if (device_ready()) return endloc;
  if (bwtgpu_bwt_from_sa_host((const uint8_t *)str, idxs, len, (uint8_t *)out, &endloc))
    report("BWT");
  return endloc;
}

int makecbwt(const char *str, int len, char *out) {
  /* histsortcomp.c:195-201 */
  int *sa = histsort(str, len);
  if (!sa) return -1;
  int e = (int)sprintcbwt(str, (unsigned int *)sa, (unsigned int)len, out);
  free(sa);
  return e;
}

int saca_makecbwt(const char *str, int len, char *out) {
  /* histsortcomp.c:204-210: same array, so same call */
  return makecbwt(str, len, out);
}

static unsigned base_at(const char *s, long i) {
  return ((unsigned char)s[i >> 2] >> (2 * (3 - (i & 3)))) & 3u;
}

void sprintbwt(char *out, const char *str, const int *bwt, int len) {
  /* histsortcomp.c:267-275: len is the length of the BWT; the last row is not printed */
  int i;
  for (i = 0; i != len; ++i) out[i] = bwt[i] ? (char)('0' + base_at(str, (long)bwt[i] - 1)) : '$';
  out[i] = 0;
}

char *makebwt(const char *str, int len) {
  /* histsortcomp.c:183-191 */
  int *sa = histsort(str, len);
  if (!sa) return NULL;
  char *buf = malloc((size_t)len + 2);
  sprintbwt(buf, str, sa, len);
  free(sa);
  return buf;
}

void putbwt(const char *str, const int *bwt, int len) {
  /* histsortcomp.c:277-287 */
  for (int i = 0; i <= len; ++i) putchar(bwt[i] ? (int)('0' + base_at(str, (long)bwt[i] - 1)) : '$');
  putchar('\n');
}

void putsg(const char *str, int idx, int len) {
  /* histsortcomp.c:252-265 */
  printf("%3d: ", idx);
  for (int i = idx; i != len; ++i) putchar((int)('0' + base_at(str, i)));
  putchar('$');
  for (int i = 0; i != idx; ++i) putchar((int)('0' + base_at(str, i)));
  putchar('\n');
}

void printseq(const char *seq, int startidx, int len) {
  /* seqindex.c:354-360 */
  for (int i = 0; i < len; ++i) putchar("ACGT"[base_at(seq, (long)startidx + i)]);
  putchar('\n');
}

/* ------------------------------------------------------------------ */
/* FM-index construction (seqindex.h)                                  */
/* ------------------------------------------------------------------ */

unsigned char *lookup_table(void) {
  /* seqindex.c:115-138: tbl[4*b + c] = how often base c occurs in byte b.
   * A constant table; kept because fm_index carries a pointer to one. */
  unsigned char *tbl = calloc(1024, 1);
  for (unsigned b = 0; b < 256; ++b)
    for (unsigned k = 0; k < 4; ++k) tbl[4 * b + ((b >> (2 * k)) & 3u)]++;
  return tbl;
}

/* rows of the reference's rank table, computed on device */
static int **rank_rows_from_device(const bwtgpu_index *ix, uint32_t len, uint32_t blocksize,
                                   int **slab_out) {
  size_t rows = 1 + ((size_t)len + blocksize - 1) / blocksize;
  int *slab = malloc(rows * 4 * sizeof *slab);
  int **index = malloc(rows * sizeof *index);
  if (!slab || !index || bwtgpu_index_rank_rows(ix, blocksize, (uint32_t *)slab)) {
    report("rank rows");
    free(slab);
    free(index);
    return NULL;
  }
  for (size_t j = 0; j < rows; ++j) index[j] = slab + 4 * j;
  if (slab_out) *slab_out = slab;
  return index;
}

int **seq_index(char *bwt, int len, int blocksize, const unsigned char *tbl) {
  /* seqindex.c:23-87.  The rows live in one allocation whose address is
   * index[0]; free(index[0]); free(index) releases them. */
  (void)tbl;
  if (device_ready()) return NULL;
  bwtgpu_index *ix = bwtgpu_index_from_parts((const uint8_t *)bwt, (uint32_t)len, 0, NULL, NULL);
  if (!ix) {
    report("seq_index");
    return NULL;
  }
  int **index = rank_rows_from_device(ix, (uint32_t)len, (uint32_t)blocksize, NULL);
  bwtgpu_index_free(ix);
  return index;
}

int seq_rank(unsigned char *bwt, int **sidx, int blocksize, int idx, char c,
             const unsigned char *lookup) {
  /* seqindex.c:91-111: reads a caller-owned table, nothing to offload */
  unsigned u = (unsigned)idx;
  int x = sidx[u / (unsigned)blocksize][(int)c];
  for (unsigned i = (u / (unsigned)blocksize) * (unsigned)blocksize / 4; i < u / 4; i++)
    x += lookup[4 * bwt[i] + c];
  for (unsigned i = 0; i < u % 4; ++i)
    if (base_at((const char *)bwt, (long)((u & ~3u) + i)) == (unsigned)c) ++x;
  return x;
}

void destroy_fmi(fm_index *fmi) {
  /* seqindex.c:140-157 */
  if (!fmi) return;
  pthread_mutex_lock(&g_reg_lock);
  struct reg_entry **pp = &g_reg, *e = NULL;
  while (*pp && (*pp)->fmi != fmi) pp = &(*pp)->next;
  if (*pp) {
    e = *pp;
    *pp = e->next;
  }
  pthread_mutex_unlock(&g_reg_lock);
  if (e) {
    bwtgpu_index_free(e->ix);
    for (int g = 1; g < FMI_MAX_SLOTS; ++g) {
      if (e->replica[g]) bwtgpu_index_free(e->replica[g]);
      pthread_mutex_destroy(&e->replica_lock[g]);
    }
    if (e->rank_slab) {
      free(e->rank_slab);
      free(fmi->rank_index);
      fmi->rank_index = NULL;
    }
    free(e);
  }
  if (fmi->rank_index) {
    /* a table made by seq_index: rows are one allocation at index[0] */
    free(fmi->rank_index[0]);
    free(fmi->rank_index);
  }
  free(fmi->bwt);
  free(fmi->idxs);
  free(fmi->lookup);
  free(fmi);
}

/* host-side fields of an fm_index whose device index exists */
static fm_index *wrap_device_index(bwtgpu_index *ix, int need_export) {
  fm_index *fmi = calloc(1, sizeof *fmi);
  uint32_t n, endloc, C[5];
  bwtgpu_index_info(ix, &n, &endloc, C);
  fmi->len = (int)n;
  fmi->endloc = (int)endloc;
  for (int i = 0; i < 5; ++i) fmi->C[i] = (int)C[i];
  if (need_export) {
    fmi->bwt = malloc(((size_t)n + 3) / 4 + 1);
    fmi->idxs = malloc(((size_t)n / 32 + 1) * sizeof(int));
    if (bwtgpu_index_export(ix, (uint8_t *)fmi->bwt, (uint32_t *)fmi->idxs)) {
      report("index export");
      free(fmi->bwt);
      free(fmi->idxs);
      free(fmi);
      return NULL;
    }
  }
  return fmi;
}

static void attach_tables(fm_index *fmi, bwtgpu_index *ix) {
  int *slab = NULL;
  fmi->lookup = lookup_table();
  /* above 2^28 bases the reference's own rows are wrong (seqindex.c:78) and
   * the table would take gigabytes of host memory; the device blocks serve */
  if ((uint32_t)fmi->len < (1u << 28))
    fmi->rank_index = rank_rows_from_device(ix, (uint32_t)fmi->len, 16, &slab);
  reg_put(fmi, ix, slab);
}

fm_index *make_fmi(const char *str, int len) {
  /* seqindex.c:161-189 */
  if (device_ready()) return NULL;
  bwtgpu_index *ix = bwtgpu_index_build((const uint8_t *)str, (uint32_t)len, NULL);
  if (!ix) {
    report("make_fmi");
    return NULL;
  }
  fm_index *fmi = wrap_device_index(ix, 1);
  if (!fmi) {
    bwtgpu_index_free(ix);
    return NULL;
  }
  attach_tables(fmi, ix);
  return fmi;
}

fm_index *make_fmi_sacak(const char *str, int len) {
  /* seqindex.c:192-212: differs from make_fmi only in the SA routine, and the
   * device build already takes any input */
  return make_fmi(str, len);
}

/* ------------------------------------------------------------------ */
/* index file (fileio.h)                                               */
/* ------------------------------------------------------------------ */

void write_index(const fm_index *fmi, FILE *f) {
  /* fileio.c:8-18 */
  uint32_t n = (uint32_t)fmi->len;
  fwrite(&fmi->len, sizeof(int), 1, f);
  fwrite(fmi->C, sizeof(int), 5, f);
  fwrite(&fmi->endloc, sizeof(int), 1, f);
  fwrite(fmi->idxs, sizeof(int), 1 + (size_t)n / 32, f);
  fwrite(fmi->bwt, 1, ((size_t)n + 3) / 4, f);
}

fm_index *read_index(const char *seq, FILE *f) {
  /* fileio.c:24-65; `seq` is unused there too */
  (void)seq;
  int err = 0;
  fm_index *fmi = calloc(1, sizeof *fmi);
  if (fread(&fmi->len, sizeof(int), 1, f) != 1) err = 1;
  if (!err && fread(fmi->C, sizeof(int), 5, f) != 5) err = 1;
  if (!err && fread(&fmi->endloc, sizeof(int), 1, f) != 1) err = 1;
  if (!err) {
    uint32_t n = (uint32_t)fmi->len;
    size_t ns = 1 + (size_t)n / 32, nb = ((size_t)n + 3) / 4;
    fmi->idxs = malloc(ns * sizeof(int));
    fmi->bwt = malloc(nb + 1);
    if (!fmi->idxs || !fmi->bwt || fread(fmi->idxs, sizeof(int), ns, f) != ns ||
        fread(fmi->bwt, 1, nb, f) != nb)
      err = 1;
  }
  if (err) {
    fprintf(stderr, "Error reading index from file\n");
    free(fmi->bwt);
    free(fmi->idxs);
    free(fmi);
    return NULL;
  }
  /* the rank table is not stored; the reference rebuilds it on the host
   * (fileio.c:62-63), here the device rebuilds its blocks */
  bwtgpu_index *ix = NULL;
  if (!device_ready())
    ix = bwtgpu_index_from_parts((const uint8_t *)fmi->bwt, (uint32_t)fmi->len,
                                 (uint32_t)fmi->endloc, (const uint32_t *)fmi->idxs, NULL);
  if (!ix) {
    report("read_index");
    free(fmi->bwt);
    free(fmi->idxs);
    free(fmi);
    return NULL;
  }
  attach_tables(fmi, ix);
  return fmi;
}

/* ------------------------------------------------------------------ */
/* single queries (seqindex.h): on the host, from the arrays of fm_index */
/* ------------------------------------------------------------------ */

/* One query per call is latency, not throughput: see the head of this file. */
static int scalars_on_host(const fm_index *fmi) {
  if (!fmi || !fmi->bwt || !fmi->idxs || !fmi->rank_index) return 0;
  const char *env = getenv("BWT_MI355X_SCALAR");
  return !(env && env[0] == 'd');
}

/* How many of the first k (0..16) bases of a 16-base group are c.  w holds the
 * group as the packed text does, first base in the top two bits. */
static inline unsigned count_in_group(uint32_t w, unsigned c, unsigned k) {
  uint32_t x = w ^ (c * 0x55555555u);        /* a base equal to c becomes 00 */
  uint32_t hit = ~(x | (x >> 1)) & 0x55555555u; /* one bit per matching base */
  if (k < 16) hit &= ~(0xFFFFFFFFu >> (2 * k)); /* keep the first k bases */
  return (unsigned)__builtin_popcount(hit);
}

static inline uint32_t group_at(const unsigned char *bwt, uint32_t first_base, uint32_t nbases) {
  /* the four bytes that hold bases first_base .. first_base + 15, zero past the end of the BWT */
  const size_t nbytes = ((size_t)nbases + 3) / 4, at = first_base / 4;
  uint32_t w = 0;
  for (unsigned k = 0; k < 4; ++k) w = (w << 8) | (at + k < nbytes ? bwt[at + k] : 0u);
  return w;
}

/* C-less Occ(c, j): occurrences of c among BWT bases [0, j), j in [0, len]: the row
 * of the 16-base block (seq_index layout, seqindex.c:23-87) plus a popcount over
 * what lies between the block's start and j */
static inline uint32_t host_occ(const fm_index *f, unsigned c, uint32_t j) {
  const uint32_t blk = j / 16;
  uint32_t x = (uint32_t)f->rank_index[blk][c];
  const unsigned k = j % 16;
  if (k) x += count_in_group(group_at((const unsigned char *)f->bwt, blk * 16, (uint32_t)f->len), c, k);
  return x;
}

static inline int host_rank(const fm_index *f, unsigned c, uint32_t idx) {
  /* seqindex.c:221-225: rows past the unstored $ row sit one place earlier in the BWT */
  const uint32_t n = (uint32_t)f->len;
  if (idx > (uint32_t)f->endloc) --idx;
  if (idx > n) return -1;
  return (int)host_occ(f, c & 3u, idx);
}

static inline int host_lf(const fm_index *f, uint32_t idx) {
  /* seqindex.c:214-219 */
  const uint32_t n = (uint32_t)f->len;
  if (idx > n) return -1;
  if (idx == (uint32_t)f->endloc) return 0;
  const uint32_t j = idx - (idx > (uint32_t)f->endloc);
  const unsigned c = base_at(f->bwt, (long)j);
  return (int)((uint32_t)f->C[c] + host_occ(f, c, j));
}

static int host_unc_sa(const fm_index *f, uint32_t idx) {
  /* seqindex.c:244-255: walk LF to the next stored row (every 32nd).  A walk on a
   * sound index ends within 31 steps; the bound keeps a damaged one from spinning. */
  const uint32_t n = (uint32_t)f->len;
  if (idx > n) return -1;
  uint32_t steps = 0;
  while (idx & 31u) {
    const int nxt = host_lf(f, idx);
    if (nxt < 0 || ++steps > n) return -1;
    idx = (uint32_t)nxt;
  }
  uint64_t x = (uint64_t)(uint32_t)f->idxs[idx / 32] + steps;
  if (x > n) x -= (uint64_t)n + 1;
  return (int)(uint32_t)x;
}

/* the loop of loc_search (seqindex.c:280-309); returns 1 when the interval emptied
 * before the last step */
static int host_interval(const fm_index *f, const char *pattern, int len, uint32_t *sp_out, uint32_t *ep_out) {
  unsigned c = (unsigned char)pattern[len - 1] & 3u;
  uint32_t sp = (uint32_t)f->C[c], ep = (uint32_t)f->C[c + 1];
  int early = 0;
  for (int i = len - 2; i >= 0; --i) {
    if (ep <= sp) {
      early = 1;
      break;
    }
    c = (unsigned char)pattern[i] & 3u;
    sp = (uint32_t)f->C[c] + (uint32_t)host_rank(f, c, sp);
    ep = (uint32_t)f->C[c] + (uint32_t)host_rank(f, c, ep);
  }
  *sp_out = sp;
  *ep_out = ep;
  return early;
}

static int host_mms(const fm_index *f, const char *pattern, int len, int *sp_out, int *ep_out) {
  /* seqindex.c:314-351: 5 is N.  Trailing Ns are skipped (:317-320); an inner N
   * is read as the base with the most occurrences in the current interval, the
   * lowest base on ties (:330-340).  A pattern of Ns only returns its length
   * and an empty interval (the reference reads before the buffer there). */
  int skips = 0;
  while (len > 0 && pattern[len - 1] == 5) {
    --len;
    ++skips;
  }
  if (len == 0) {
    *sp_out = 0;
    *ep_out = 0;
    return skips;
  }
  unsigned c = (unsigned char)pattern[len - 1] & 3u;
  uint32_t start = (uint32_t)f->C[c], end = (uint32_t)f->C[c + 1], sp = start, ep = end;
  int i = len - 2;
  for (; i >= 0; --i) {
    if (end <= start) break;
    sp = start;
    ep = end;
    if (pattern[i] == 5) {
      int64_t best = -1;
      uint32_t bs = 0, be = 0;
      for (unsigned d = 0; d < 4; ++d) {
        const uint32_t a = (uint32_t)f->C[d] + (uint32_t)host_rank(f, d, start);
        const uint32_t b = (uint32_t)f->C[d] + (uint32_t)host_rank(f, d, end);
        if ((int64_t)b - (int64_t)a > best) {
          best = (int64_t)b - (int64_t)a;
          bs = a;
          be = b;
        }
      }
      start = bs;
      end = be;
    } else {
      c = (unsigned char)pattern[i] & 3u;
      start = (uint32_t)f->C[c] + (uint32_t)host_rank(f, c, start);
      end = (uint32_t)f->C[c] + (uint32_t)host_rank(f, c, end);
    }
  }
  if (end <= start) {
    *sp_out = (int)sp;
    *ep_out = (int)ep;
    return len - i - 2 + skips;
  }
  *sp_out = (int)start;
  *ep_out = (int)end;
  return len - i - 1 + skips;
}

int rank(const fm_index *fmi, char c, int idx) {
  if (scalars_on_host(fmi)) return host_rank(fmi, (unsigned char)c, (uint32_t)idx);
  bwtgpu_index *ix = device_index(fmi);
  uint8_t cc = (uint8_t)c;
  uint32_t i = (uint32_t)idx, out = 0;
  if (!ix || bwtgpu_occ_host(ix, &cc, &i, 1, 1, &out)) {
    if (ix) report("rank");
    return -1;
  }
  return (int)out;
}

int lf(const fm_index *fmi, int idx) {
  if (scalars_on_host(fmi)) return host_lf(fmi, (uint32_t)idx);
  bwtgpu_index *ix = device_index(fmi);
  uint32_t i = (uint32_t)idx, out = 0;
  if (!ix || bwtgpu_lf_host(ix, &i, 1, &out)) {
    if (ix) report("lf");
    return -1;
  }
  return (int)out;
}

int unc_sa(const fm_index *fmi, int idx) {
  if (scalars_on_host(fmi)) return host_unc_sa(fmi, (uint32_t)idx);
  int pos = -1;
  fmi_unc_sa_batch(fmi, &idx, 1, &pos);
  return pos;
}

void loc_search(const fm_index *fmi, const char *pattern, int len, int *sp, int *ep) {
  if (len > 0 && scalars_on_host(fmi)) {
    uint32_t s, e;
    host_interval(fmi, pattern, len, &s, &e);
    *sp = (int)s;
    *ep = (int)e;
    return;
  }
  fmi_search_batch(fmi, pattern, len, len, 1, sp, ep);
}

int reverse_search(const fm_index *fmi, const char *pattern, int len) {
  if (len > 0 && scalars_on_host(fmi)) {
    /* seqindex.c:228-242: 0 if the interval emptied before the last step, else ep - sp + 1 (sic) */
    uint32_t s, e;
    return host_interval(fmi, pattern, len, &s, &e) ? 0 : (int)(e - s + 1u);
  }
  int r = 0;
  fmi_reverse_search_batch(fmi, pattern, len, len, 1, &r);
  return r;
}

int locate(const fm_index *fmi, const char *pattern, int len) {
  if (len > 0 && scalars_on_host(fmi)) {
    uint32_t s, e;
    if (host_interval(fmi, pattern, len, &s, &e)) return -1; /* seqindex.c:268-270 */
    if (e - s != 1) printf("Warning: multiple matches found (returned first)\n"); /* seqindex.c:274-275 */
    return host_unc_sa(fmi, s); /* -1 for s = len + 1, where the reference reads out of bounds */
  }
  bwtgpu_index *ix = device_index(fmi);
  uint32_t sp, ep, row, pos;
  if (!ix || bwtgpu_search_host(ix, (const uint8_t *)pattern, (uint32_t)len, (uint64_t)len, 1, &sp, &ep,
                                &row, &pos)) {
    if (ix) report("locate");
    return -1;
  }
  if (row == 0xFFFFFFFFu) return -1; /* seqindex.c:268-270 */
  if (ep - sp != 1) printf("Warning: multiple matches found (returned first)\n"); /* seqindex.c:274-275 */
  return (int)pos;
}

/* ------------------------------------------------------------------ */
/* batch forms                                                         */
/* ------------------------------------------------------------------ */

struct search_ctx {
  const char *pats;
  const unsigned *words;
  int len;
  long stride;
  int *sp, *ep, *pos, *result;
};
#define AT(p, lo) ((p) ? (p) + (lo) : NULL)

static int share_search(bwtgpu_index *ix, long lo, long hi, void *vc) {
  const struct search_ctx *c = vc;
  if (bwtgpu_search_host(ix, (const uint8_t *)c->pats + lo * c->stride, (uint32_t)c->len, (uint64_t)c->stride,
                         (uint64_t)(hi - lo), (uint32_t *)AT(c->sp, lo), (uint32_t *)AT(c->ep, lo), NULL,
                         (uint32_t *)AT(c->pos, lo))) {
    report("search");
    return -1;
  }
  return 0;
}

static int share_reverse_search(bwtgpu_index *ix, long lo, long hi, void *vc) {
  const struct search_ctx *c = vc;
  /* seqindex.c:233-241, evaluated on the device: one int per read comes back */
  if (bwtgpu_reverse_search_host(ix, (const uint8_t *)c->pats + lo * c->stride, (uint32_t)c->len,
                                 (uint64_t)c->stride, (uint64_t)(hi - lo), (int32_t *)c->result + lo)) {
    report("reverse_search");
    return -1;
  }
  return 0;
}

static int share_search_packed(bwtgpu_index *ix, long lo, long hi, void *vc) {
  const struct search_ctx *c = vc;
  const long words = (c->len + 15) / 16;
  if (bwtgpu_search_packed_host(ix, c->words + lo * words, (uint32_t)c->len, (uint64_t)(hi - lo),
                                (uint32_t *)AT(c->sp, lo), (uint32_t *)AT(c->ep, lo), NULL,
                                (uint32_t *)AT(c->pos, lo))) {
    report("packed search");
    return -1;
  }
  return 0;
}

int fmi_search_batch(const fm_index *fmi, const char *pats, int len, long stride, long nreads,
                     int *sp, int *ep) {
  struct search_ctx c = {.pats = pats, .len = len, .stride = stride, .sp = sp, .ep = ep};
  return run_shared(fmi, nreads, share_search, &c);
}

int fmi_reverse_search_batch(const fm_index *fmi, const char *pats, int len, long stride,
                             long nreads, int *result) {
  struct search_ctx c = {.pats = pats, .len = len, .stride = stride, .result = result};
  return run_shared(fmi, nreads, share_reverse_search, &c);
}

int fmi_locate_batch(const fm_index *fmi, const char *pats, int len, long stride, long nreads,
                     int *pos) {
  struct search_ctx c = {.pats = pats, .len = len, .stride = stride, .pos = pos};
  return run_shared(fmi, nreads, share_search, &c); /* 0xFFFFFFFF reads as -1 */
}

int fmi_search_locate_batch(const fm_index *fmi, const char *pats, int len, long stride, long nreads,
                            int *sp, int *ep, int *pos) {
  /* one pass over the reads: an index with position records has the positions by the time it has the intervals */
  struct search_ctx c = {.pats = pats, .len = len, .stride = stride, .sp = sp, .ep = ep, .pos = pos};
  return run_shared(fmi, nreads, share_search, &c);
}

void fmi_pack_reads(const char *pats, int len, long stride, long nreads, unsigned *words) {
  /* host threads only: no device is needed to pack */
  if (bwtgpu_pack_reads_host((const uint8_t *)pats, (uint32_t)len, (uint64_t)stride, (uint64_t)nreads, words))
    report("pack reads");
}

int fmi_search_batch_packed(const fm_index *fmi, const unsigned *words, int len, long nreads,
                            int *sp, int *ep, int *pos) {
  struct search_ctx c = {.words = words, .len = len, .sp = sp, .ep = ep, .pos = pos};
  return run_shared(fmi, nreads, share_search_packed, &c);
}

struct rows_ctx {
  const int *rows;
  int *pos;
};

static int share_unc_sa(bwtgpu_index *ix, long lo, long hi, void *vc) {
  const struct rows_ctx *c = vc;
  if (bwtgpu_locate_rows_host(ix, (const uint32_t *)c->rows + lo, (uint64_t)(hi - lo), (uint32_t *)c->pos + lo)) {
    report("unc_sa");
    return -1;
  }
  return 0;
}

int fmi_unc_sa_batch(const fm_index *fmi, const int *rows, long nrows, int *pos) {
  struct rows_ctx c = {rows, pos};
  return run_shared(fmi, nrows, share_unc_sa, &c);
}

struct mms_ctx {
  const char *pats;
  int len;
  long stride;
  int *matched, *sp, *ep;
};

static int share_mms(bwtgpu_index *ix, long lo, long hi, void *vc) {
  const struct mms_ctx *c = vc;
  if (bwtgpu_mms_host(ix, (const uint8_t *)c->pats + lo * c->stride, (uint32_t)c->len, (uint64_t)c->stride,
                      (uint64_t)(hi - lo), c->matched + lo, (uint32_t *)c->sp + lo, (uint32_t *)c->ep + lo)) {
    report("mms");
    return -1;
  }
  return 0;
}

int fmi_mms_batch(const fm_index *fmi, const char *pats, int len, long stride, long nreads,
                  int *matched, int *sp, int *ep) {
  struct mms_ctx c = {pats, len, stride, matched, sp, ep};
  return run_shared(fmi, nreads, share_mms, &c);
}

int mms(const fm_index *fmi, const char *pattern, int len, int *sp, int *ep) {
  if (len > 0 && scalars_on_host(fmi)) return host_mms(fmi, pattern, len, sp, ep);
  int matched = 0;
  if (fmi_mms_batch(fmi, pattern, len, len, 1, &matched, sp, ep)) return 0;
  return matched;
}

int fmi_search_reads_batch(const fm_index *fmi, const char *reads, const unsigned long *offsets,
                           long nreads, int anchor, int *out) {
  bwtgpu_index *ix = device_index(fmi);
  if (!ix) return -1;
  if (bwtgpu_search_reads_host(ix, (const uint8_t *)reads, (const uint64_t *)offsets, (uint64_t)nreads,
                               anchor, (uint32_t *)out)) {
    report("search_reads");
    return -1;
  }
  return 0;
}

int fmi_align_reads_batch(const fm_index *fmi, const char *seq, const char *reads,
                          const unsigned long *offsets, long nreads, int anchor_len, int *pos,
                          char *ops, unsigned *nops) {
  bwtgpu_index *ix = device_index(fmi);
  if (!ix) return -1;
  if (bwtgpu_align_reads_host(ix, (const uint8_t *)seq, (const uint8_t *)reads, (const uint64_t *)offsets,
                              (uint64_t)nreads, anchor_len, pos, (uint8_t *)ops, nops)) {
    report("align_reads");
    return -1;
  }
  return 0;
}

void fmi_encode_read(const char *ascii, long len, char *out) {
  /* single_align.c:450-474 */
  for (long i = 0; i < len; ++i) {
    switch (ascii[i]) {
      case 'A': out[i] = 0; break;
      case 'C': out[i] = 1; break;
      case 'G': out[i] = 2; break;
      case 'T': out[i] = 3; break;
      default: out[i] = 5;
    }
  }
}

/* ------------------------------------------------------------------ */
/* genome packer                                                       */
/* ------------------------------------------------------------------ */

int fmi_attach_text(const fm_index *fmi, const char *seq) {
  bwtgpu_index *ix = device_index(fmi);
  if (!ix || !seq) return -1;
  if (bwtgpu_index_attach_text(ix, (const uint8_t *)seq, NULL)) {
    report("attach text");
    return -1;
  }
  return 0;
}

void fmi_pack_ascii(const char *ascii, long len, char *out) {
  /* build_index.c:54-77: exact-case C, G, T; everything else, and everything
   * past the end of the file, is A */
  static const unsigned char code[256] = {['C'] = 1, ['G'] = 2, ['T'] = 3};
  long full = len / 4;
  for (long i = 0; i < full; ++i) {
    const unsigned char *p = (const unsigned char *)ascii + 4 * i;
    out[i] = (char)(code[p[0]] << 6 | code[p[1]] << 4 | code[p[2]] << 2 | code[p[3]]);
  }
  /* build_index.c:80-87: the tail loop's test is (i < len) & 3 and its reads
   * all hit end of file, so it leaves the byte of the last len%4 bases zero */
  out[full] = 0;
}
