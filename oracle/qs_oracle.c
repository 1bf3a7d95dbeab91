/*
 * qs_oracle.c -- CPU restatement of the QuartetScores hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (quartetscores_amd/, the
 * C-ABI library, the CLI) may include, link, dlopen or execute this file.  It
 * is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / reported CPU baseline, never as the thing shipped.
 *
 * PARITY STATUS: "parity unpinned" at the genesis boundary.
 *   - The reference holds NO tests, golden vectors or fixtures for this path
 *     (SURVEY.md section 4), and the reference program cannot be built in this
 *     image: QuartetCounterLookup.hpp / QuartetScoreComputer.hpp /
 *     TreeInformation.hpp include genesis v0.16.0 (un-vendored submodule,
 *     cmake/GenesisDownload.cmake:47), which is absent.
 *   - The ONE reference source on the path that compiles from its own file,
 *     src/quartet_lookup_table.hpp, IS built unmodified into
 *     oracle/_ref/libqs_reftable.so (oracle/Makefile) and this file's
 *     rank / slot restatement is checked against it exhaustively
 *     (tests/test_oracle_reftable.py).
 *   - Known-answer vectors D1..D6 of SURVEY.md Appendix D (produced by the
 *     surveyor from the unmodified reference headers) are committed under
 *     tests/golden/ and this oracle reproduces them (tests/test_oracle_golden.py).
 *   - An independent split-based brute-force counter (tests/bruteforce.py)
 *     agrees with the counts.
 *
 * What is restated, with the reference lines each function follows
 * (paths relative to /root/reference/src):
 *   tree model + eulertour      SURVEY.md Appendix B/E (genesis conventions)
 *   qso_rank / qso_slot         quartet_lookup_table.hpp:170-212,141-168,87-111
 *   update_three_clades         QuartetCounterLookup.hpp:65-106
 *   subtree_leaf_indices        QuartetCounterLookup.hpp:116-121
 *   update_three_links          QuartetCounterLookup.hpp:134-154
 *   update_quartets             QuartetCounterLookup.hpp:166-188
 *   qso_count (countQuartets)   QuartetCounterLookup.hpp:196-238, 245-273
 *   lookup_quartet_count        QuartetCounterLookup.hpp:282-290
 *   count_quartet_occurrences   QuartetCounterLookup.hpp:299-318
 *   tree_information_*          TreeInformation.hpp:40-113
 *   log_score                   QuartetScoreComputer.hpp:135-159
 *   get_path_inner_links        QuartetScoreComputer.hpp:169-194
 *   process_node_pair           QuartetScoreComputer.hpp:379-490
 *   scores_bifurcating          QuartetScoreComputer.hpp:495-508
 *   scores_multifurcating       QuartetScoreComputer.hpp:513-593
 *   qso_raw_qic                 QuartetScoreComputer.hpp:623-690
 *   CINT width selection        QuartetScores.cpp:115-147
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp, like CMakeLists.txt:42,71).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* Tree model (genesis-like: nodes, links with next/outer, edges)             */
/* ------------------------------------------------------------------------- */

typedef struct { int next, outer, node, edge; } Link;
typedef struct { int link; int parent; int depth; int name; int nchild; } Node;

typedef struct {
    int n_nodes, n_links, n_edges, root;
    Node *nodes;
    Link *links;
    int *edge_primary;   /* edge -> link on the root side  */
    int *edge_secondary; /* edge -> link on the child side (the child's primary link) */
    char **names;        /* per node, NULL for unnamed */
} Tree;

typedef struct TmpNode {
    int first_child, last_child, next_sibling, parent;
    char *name;
} TmpNode;

typedef struct {
    const char *s;
    size_t pos, len;
    TmpNode *tn;
    int n_tn, cap_tn;
    char err[256];
} Parser;

static int tmp_new(Parser *p, int parent) {
    if (p->n_tn == p->cap_tn) {
        p->cap_tn = p->cap_tn ? p->cap_tn * 2 : 64;
        p->tn = (TmpNode *)realloc(p->tn, sizeof(TmpNode) * (size_t)p->cap_tn);
    }
    int id = p->n_tn++;
    p->tn[id].first_child = p->tn[id].last_child = p->tn[id].next_sibling = -1;
    p->tn[id].parent = parent;
    p->tn[id].name = NULL;
    if (parent >= 0) {
        if (p->tn[parent].last_child < 0) p->tn[parent].first_child = id;
        else p->tn[p->tn[parent].last_child].next_sibling = id;
        p->tn[parent].last_child = id;
    }
    return id;
}

static void skip_ws_comments(Parser *p) {
    for (;;) {
        while (p->pos < p->len && (p->s[p->pos] == ' ' || p->s[p->pos] == '\t' || p->s[p->pos] == '\n' ||
                                   p->s[p->pos] == '\r'))
            p->pos++;
        if (p->pos < p->len && p->s[p->pos] == '[') {
            while (p->pos < p->len && p->s[p->pos] != ']') p->pos++;
            if (p->pos < p->len) p->pos++;
            continue;
        }
        break;
    }
}

static char *parse_label(Parser *p) {
    skip_ws_comments(p);
    if (p->pos >= p->len) return NULL;
    size_t cap = 32, n = 0;
    char *buf = (char *)malloc(cap);
    if (p->s[p->pos] == '\'') {
        p->pos++;
        while (p->pos < p->len) {
            char c = p->s[p->pos];
            if (c == '\'') {
                if (p->pos + 1 < p->len && p->s[p->pos + 1] == '\'') { p->pos += 2; c = '\''; }
                else { p->pos++; break; }
            } else p->pos++;
            if (n + 2 > cap) { cap *= 2; buf = (char *)realloc(buf, cap); }
            buf[n++] = c;
        }
    } else {
        while (p->pos < p->len) {
            char c = p->s[p->pos];
            if (c == '(' || c == ')' || c == ',' || c == ':' || c == ';' || c == '[' || c == ' ' || c == '\t' ||
                c == '\n' || c == '\r')
                break;
            if (n + 2 > cap) { cap *= 2; buf = (char *)realloc(buf, cap); }
            buf[n++] = (c == '_') ? '_' : c;
            p->pos++;
        }
    }
    buf[n] = 0;
    if (n == 0) { free(buf); return NULL; }
    return buf;
}

static void skip_branch_info(Parser *p) {
    skip_ws_comments(p);
    while (p->pos < p->len && p->s[p->pos] == ':') {
        p->pos++;
        skip_ws_comments(p);
        while (p->pos < p->len) {
            char c = p->s[p->pos];
            if ((c >= '0' && c <= '9') || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E') p->pos++;
            else break;
        }
        skip_ws_comments(p);
    }
}

static int parse_subtree(Parser *p, int parent) {
    skip_ws_comments(p);
    int id = tmp_new(p, parent);
    if (p->pos < p->len && p->s[p->pos] == '(') {
        p->pos++;
        for (;;) {
            if (parse_subtree(p, id) < 0) return -1;
            skip_ws_comments(p);
            if (p->pos < p->len && p->s[p->pos] == ',') { p->pos++; continue; }
            if (p->pos < p->len && p->s[p->pos] == ')') { p->pos++; break; }
            snprintf(p->err, sizeof p->err, "newick: expected ',' or ')' at offset %zu", p->pos);
            return -1;
        }
    }
    p->tn[id].name = parse_label(p);
    skip_branch_info(p);
    return id;
}

static void tree_free(Tree *t) {
    if (!t) return;
    if (t->names) for (int i = 0; i < t->n_nodes; i++) free(t->names[i]);
    free(t->names); free(t->nodes); free(t->links); free(t->edge_primary); free(t->edge_secondary);
    free(t);
}

/* Nodes numbered in preorder; all of a node's links are allocated before its
 * children are visited. */
static Tree *tree_from_tmp(Parser *p) {
    int nn = p->n_tn;
    Tree *t = (Tree *)calloc(1, sizeof(Tree));
    t->n_nodes = nn;
    t->n_edges = nn - 1;
    t->n_links = 2 * (nn - 1);
    t->nodes = (Node *)calloc((size_t)nn, sizeof(Node));
    t->links = (Link *)calloc((size_t)(t->n_links > 0 ? t->n_links : 1), sizeof(Link));
    t->edge_primary = (int *)calloc((size_t)(nn > 1 ? nn - 1 : 1), sizeof(int));
    t->edge_secondary = (int *)calloc((size_t)(nn > 1 ? nn - 1 : 1), sizeof(int));
    t->names = (char **)calloc((size_t)nn, sizeof(char *));
    t->root = 0;
    /* tmp ids are already preorder (parse order) */
    int next_link = 0, next_edge = 0;
    /* explicit stack, preorder */
    int *stack = (int *)malloc(sizeof(int) * (size_t)nn);
    int *uplink = (int *)malloc(sizeof(int) * (size_t)nn); /* parent's child link for each node */
    int sp = 0;
    stack[sp++] = 0;
    uplink[0] = -1;
    while (sp > 0) {
        int x = stack[--sp];
        TmpNode *tx = &p->tn[x];
        int nchild = 0;
        for (int c = tx->first_child; c >= 0; c = p->tn[c].next_sibling) nchild++;
        t->nodes[x].nchild = nchild;
        t->nodes[x].parent = tx->parent;
        t->nodes[x].depth = tx->parent >= 0 ? t->nodes[tx->parent].depth + 1 : 0;
        t->names[x] = tx->name; tx->name = NULL;
        int nl = nchild + (tx->parent >= 0 ? 1 : 0);
        int first = next_link;
        next_link += nl;
        for (int i = 0; i < nl; i++) {
            t->links[first + i].node = x;
            t->links[first + i].next = first + (i + 1) % nl;
        }
        t->nodes[x].link = first;
        int li = first;
        if (tx->parent >= 0) {
            int pl = uplink[x];
            int e = next_edge++;
            t->links[first].outer = pl;
            t->links[pl].outer = first;
            t->links[first].edge = e;
            t->links[pl].edge = e;
            t->edge_primary[e] = pl;
            t->edge_secondary[e] = first;
            li = first + 1;
        }
        /* children get this node's child links in order; push in reverse so that
         * the first child is popped (numbered) first -> preorder */
        int k = 0;
        for (int c = tx->first_child; c >= 0; c = p->tn[c].next_sibling) uplink[c] = li + (k++);
        int *tmp = (int *)malloc(sizeof(int) * (size_t)(nchild > 0 ? nchild : 1));
        k = 0;
        for (int c = tx->first_child; c >= 0; c = p->tn[c].next_sibling) tmp[k++] = c;
        for (int i = nchild - 1; i >= 0; i--) stack[sp++] = tmp[i];
        free(tmp);
    }
    free(stack); free(uplink);
    return t;
}

/* NOTE: tmp ids are created in parse order = preorder, but the explicit stack
 * above assigns LINK and EDGE numbers in preorder as well because children are
 * popped first-child-first. Node index == tmp id. */

static Tree *tree_parse(const char *s, size_t len, size_t *consumed, char *err, int errlen) {
    Parser p; memset(&p, 0, sizeof p);
    p.s = s; p.len = len; p.pos = 0;
    skip_ws_comments(&p);
    if (p.pos >= p.len) { if (consumed) *consumed = p.pos; return NULL; }
    int r = parse_subtree(&p, -1);
    if (r < 0) {
        if (err) snprintf(err, (size_t)errlen, "%s", p.err);
        for (int i = 0; i < p.n_tn; i++) free(p.tn[i].name);
        free(p.tn);
        if (consumed) *consumed = len;
        return NULL;
    }
    skip_ws_comments(&p);
    if (p.pos < p.len && p.s[p.pos] == ';') p.pos++;
    Tree *t = tree_from_tmp(&p);
    for (int i = 0; i < p.n_tn; i++) free(p.tn[i].name);
    free(p.tn);
    if (consumed) *consumed = p.pos;
    return t;
}

static inline int node_is_leaf(const Tree *t, int x) { return t->nodes[x].nchild == 0; }
static inline int node_primary_link(const Tree *t, int x) { return t->nodes[x].link; }

/* eulertour: start at the root's link, step link = link.outer().next(), every
 * link visited once. Returns the link sequence. */
static int *eulertour_links(const Tree *t, int *out_len) {
    int *seq = (int *)malloc(sizeof(int) * (size_t)(t->n_links > 0 ? t->n_links : 1));
    int n = 0;
    if (t->n_links == 0) { *out_len = 0; return seq; }
    int start = t->nodes[t->root].link, cur = start;
    do {
        seq[n++] = cur;
        cur = t->links[t->links[cur].outer].next;
    } while (cur != start && n < t->n_links);
    *out_len = n;
    return seq;
}

/* ------------------------------------------------------------------------- */
/* Compact table index + slot (quartet_lookup_table.hpp)                       */
/* ------------------------------------------------------------------------- */

/* quartet_lookup_table.hpp:170-212 (sort) + :141-168 (binomial sum) */
uint64_t qso_rank(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    uint64_t ta, tb, tc, td, low1, high1, low2, high2, middle1, middle2;
    if (a < b) { low1 = a; high1 = b; } else { low1 = b; high1 = a; }
    if (c < d) { low2 = c; high2 = d; } else { low2 = d; high2 = c; }
    if (low1 < low2) { td = low1; middle1 = low2; } else { td = low2; middle1 = low1; }
    if (high1 > high2) { ta = high1; middle2 = high2; } else { ta = high2; middle2 = high1; }
    if (middle1 < middle2) { tc = middle1; tb = middle2; } else { tc = middle2; tb = middle1; }
    uint64_t res = 0;
    res += (ta * (ta - 1) * (ta - 2) * (ta - 3)) / 24;
    res += (tb * (tb - 1) * (tb - 2)) / 6;
    res += (tc * (tc - 1)) / 2;
    res += td;
    return res;
}

/* quartet_lookup_table.hpp:87-111 */
int qso_slot(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    int ac = (a < c), ad = (a < d), bc = (b < c), bd = (b < d);
    int x = ((ac) & (ad) & (bc) & (bd)) | ((!ac) & (!bc) & (!ad) & (!bd));
    int ab_in_cd = ((!ac) & (ad) & (!bc) & (bd)) | ((!ad) & (ac) & (!bd) & (bc));
    int cd_in_ab = ((ac) & (!bc) & (ad) & (!bd)) | ((bc) & (!ac) & (bd) & (!ad));
    int z = ab_in_cd | cd_in_ab;
    int y = !x & !z;
    return y + 2 * z;
}

/* ------------------------------------------------------------------------- */
/* Oracle handle                                                               */
/* ------------------------------------------------------------------------- */

typedef struct {
    Tree *ref;
    int n;                 /* taxa */
    int *ref_id_to_lookup; /* ref node idx -> lookup id (QCL:252-258) */
    int *lookup_to_node;   /* inverse */
    /* count tables */
    int savemem, bits;
    uint64_t mask;
    uint64_t nq;
    void *fast;    /* n^4 cells of CINT  */
    void *compact; /* C(n,4)*3 cells of CINT */
    uint64_t m;    /* number of evaluation trees counted */
    /* reference-tree info for scoring (QSC:710-717) */
    size_t *eulerTourLeaves; /* ref node indices */
    int n_etl;
    int *linkToEulerLeafIndex;
    /* TreeInformation */
    int *ti_tour_nodes; int ti_len; int *ti_levels; int *ti_first; int *ti_sparse; int ti_logn;
    int *dist_to_root;
    /* scores */
    double *LQ, *QP, *EQP;
    int have_qp;
    int qp_exact64; /* 0 = reference-compatible 32-bit wrap (QSC:382), 1 = 64-bit sums */
    double t_count, t_score;
    /* bench.py's cpu_baseline leg only: stop counting after budget_s seconds (checked per inner node) and
     * report how many table increments were done; 0 = off (every parity test) */
    double budget_s, budget_t0;
    volatile int budget_hit;
    /* quartet_lookup_table.hpp:79-85: the const get_tuple throws std::runtime_error when the index of the (sorted) ids
     * lies behind the table; the reference does not catch it (the scoring loop dies). Restated as a sticky flag + the
     * exception's text; qso_score returns -3. Reached only with REPEATED ids: savemem + a degree-2 reference root. */
    volatile int threw;
    char threw_what[128];
    unsigned long long incr_done;
    int prefault;
    int table_savemem, table_bits;
    char err[256];
} Oracle;

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

const char *qso_last_error(void *h) { return ((Oracle *)h)->err; }

void *qso_create(const char *ref_newick, char *err, int errlen) {
    size_t used = 0;
    Tree *t = tree_parse(ref_newick, strlen(ref_newick), &used, err, errlen);
    if (!t) { if (err && !err[0]) snprintf(err, (size_t)errlen, "empty reference tree"); return NULL; }
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    o->ref = t;
    o->ref_id_to_lookup = (int *)malloc(sizeof(int) * (size_t)t->n_nodes);
    for (int i = 0; i < t->n_nodes; i++) o->ref_id_to_lookup[i] = -1;
    o->lookup_to_node = (int *)malloc(sizeof(int) * (size_t)t->n_nodes);
    int len; int *seq = eulertour_links(t, &len);
    /* QCL:252-258 lookup ids in tour order; QSC:711-717 eulerTourLeaves / lTELI */
    o->eulerTourLeaves = (size_t *)malloc(sizeof(size_t) * (size_t)t->n_nodes);
    o->linkToEulerLeafIndex = (int *)calloc((size_t)(t->n_links > 0 ? t->n_links : 1), sizeof(int));
    int n = 0;
    for (int i = 0; i < len; i++) {
        int l = seq[i], x = t->links[l].node;
        if (node_is_leaf(t, x)) {
            o->ref_id_to_lookup[x] = n;
            o->lookup_to_node[n] = x;
            o->eulerTourLeaves[n] = (size_t)x;
            n++;
        }
        o->linkToEulerLeafIndex[l] = n;
    }
    free(seq);
    o->n = n; o->n_etl = n;
    o->nq = n >= 4 ? ((uint64_t)n * (uint64_t)(n - 1) * (uint64_t)(n - 2) * (uint64_t)(n - 3)) / 24 : 0;
    return o;
}

void qso_destroy(void *h) {
    Oracle *o = (Oracle *)h;
    if (!o) return;
    tree_free(o->ref);
    free(o->ref_id_to_lookup); free(o->lookup_to_node); free(o->fast); free(o->compact);
    free(o->eulerTourLeaves); free(o->linkToEulerLeafIndex);
    free(o->ti_tour_nodes); free(o->ti_levels); free(o->ti_first); free(o->ti_sparse); free(o->dist_to_root);
    free(o->LQ); free(o->QP); free(o->EQP);
    free(o);
}

int qso_n_taxa(void *h) { return ((Oracle *)h)->n; }
int qso_n_edges(void *h) { return ((Oracle *)h)->ref->n_edges; }
int qso_n_nodes(void *h) { return ((Oracle *)h)->ref->n_nodes; }
uint64_t qso_n_quartets(void *h) { return ((Oracle *)h)->nq; }
const char *qso_taxon_name(void *h, int lookup_id) {
    Oracle *o = (Oracle *)h;
    const char *s = o->ref->names[o->lookup_to_node[lookup_id]];
    return s ? s : "";
}
double qso_time_count(void *h) { return ((Oracle *)h)->t_count; }
/* bench.py cpu_baseline: bound the next qso_count calls to `seconds` of counting (0 = unbounded, the default) and
 * optionally touch the table's pages before the clock starts. A bounded count leaves an INCOMPLETE table. */
void qso_set_budget(void *h, double seconds, int prefault) { Oracle *o = (Oracle *)h; o->budget_s = seconds; o->prefault = prefault; }
unsigned long long qso_increments_done(void *h) { return ((Oracle *)h)->incr_done; }
double qso_time_score(void *h) { return ((Oracle *)h)->t_score; }

/* QuartetScores.cpp:115-147: CINT width follows m */
int qso_cint_bits_for_m(uint64_t m) {
    if (m < (1ull << 8)) return 8;
    if (m < (1ull << 16)) return 16;
    if (m < (1ull << 32)) return 32;
    return 64;
}

/* ------------------------------------------------------------------------- */
/* Counting (QuartetCounterLookup.hpp)                                         */
/* ------------------------------------------------------------------------- */

#define DEFINE_CLADES(NAME, T)                                                                              \
    /* QuartetCounterLookup.hpp:65-106 */                                                                   \
    static void NAME(Oracle *o, size_t sS1, size_t eS1, size_t sS2, size_t eS2, size_t sS3, size_t eS3,      \
                     const int *etl, size_t L, uint64_t mult) {                                             \
        const size_t n = (size_t)o->n, n2 = n * n, n3 = n2 * n;                                             \
        T *fast = (T *)o->fast;                                                                             \
        T *cmp = (T *)o->compact;                                                                           \
        size_t ai = sS1, bi = sS2, ci = sS3;                                                                \
        while (ai != eS1) {                                                                                 \
            size_t a = (size_t)etl[ai];                                                                     \
            size_t a2i = (ai + 1) % L;                                                                      \
            while (a2i != eS1) {                                                                            \
                size_t a2 = (size_t)etl[a2i];                                                               \
                while (bi != eS2) {                                                                         \
                    size_t b = (size_t)etl[bi];                                                             \
                    while (ci != eS3) {                                                                     \
                        size_t c = (size_t)etl[ci];                                                         \
                        if (o->savemem) {                                                                   \
                            uint64_t r = qso_rank(a, a2, b, c);                                             \
                            int s = qso_slot(a, a2, b, c);                                                  \
                            cmp[r * 3 + (uint64_t)s] = (T)(cmp[r * 3 + (uint64_t)s] + (T)mult);             \
                        } else {                                                                            \
                            fast[a * n3 + a2 * n2 + b * n + c] = (T)(fast[a * n3 + a2 * n2 + b * n + c] + (T)mult); \
                        }                                                                                   \
                        ci = (ci + 1) % L;                                                                  \
                    }                                                                                       \
                    bi = (bi + 1) % L;                                                                      \
                    ci = sS3;                                                                               \
                }                                                                                           \
                a2i = (a2i + 1) % L;                                                                        \
                bi = sS2;                                                                                   \
                ci = sS3;                                                                                   \
            }                                                                                               \
            ai = (ai + 1) % L;                                                                              \
            bi = sS2;                                                                                       \
            ci = sS3;                                                                                       \
        }                                                                                                   \
    }

DEFINE_CLADES(clades_u8, uint8_t)
DEFINE_CLADES(clades_u16, uint16_t)
DEFINE_CLADES(clades_u32, uint32_t)
DEFINE_CLADES(clades_u64, uint64_t)

static void update_three_clades(Oracle *o, size_t s1, size_t e1, size_t s2, size_t e2, size_t s3, size_t e3,
                                const int *etl, size_t L, uint64_t mult) {
    switch (o->bits) {
    case 8: clades_u8(o, s1, e1, s2, e2, s3, e3, etl, L, mult); break;
    case 16: clades_u16(o, s1, e1, s2, e2, s3, e3, etl, L, mult); break;
    case 32: clades_u32(o, s1, e1, s2, e2, s3, e3, etl, L, mult); break;
    default: clades_u64(o, s1, e1, s2, e2, s3, e3, etl, L, mult); break;
    }
}

/* QuartetCounterLookup.hpp:116-121 (note: modulo the LINK count, as written) */
static void subtree_leaf_indices(const Tree *t, int link, const int *lteli, size_t *first, size_t *second) {
    int outer = t->links[link].outer;
    size_t nl = (size_t)t->n_links;
    *first = (size_t)lteli[link] % nl;
    *second = (size_t)lteli[outer] % nl;
}

/* QuartetCounterLookup.hpp:134-154 */
static void update_three_links(Oracle *o, const Tree *t, int l1, int l2, int l3, const int *etl, size_t L,
                               const int *lteli, uint64_t mult) {
    size_t a1, b1, a2, b2, a3, b3;
    subtree_leaf_indices(t, l1, lteli, &a1, &b1);
    subtree_leaf_indices(t, l2, lteli, &a2, &b2);
    subtree_leaf_indices(t, l3, lteli, &a3, &b3);
    size_t s1 = a1 % L, e1 = b1 % L, s2 = a2 % L, e2 = b2 % L, s3 = a3 % L, e3 = b3 % L;
    if (o->budget_s > 0) { /* increments of the three calls below: C(|S1|,2)|S2||S3| + ... (trip count of QCL:73-105) */
        const unsigned long long n1 = (e1 + L - s1) % L, n2 = (e2 + L - s2) % L, n3 = (e3 + L - s3) % L;
        const unsigned long long inc = n1 * (n1 - (n1 > 0)) / 2 * n2 * n3 + n2 * (n2 - (n2 > 0)) / 2 * n1 * n3 + n3 * (n3 - (n3 > 0)) / 2 * n1 * n2;
        __atomic_fetch_add(&o->incr_done, inc, __ATOMIC_RELAXED);
    }
    update_three_clades(o, s1, e1, s2, e2, s3, e3, etl, L, mult);
    update_three_clades(o, s2, e2, s1, e1, s3, e3, etl, L, mult);
    update_three_clades(o, s3, e3, s1, e1, s2, e2, etl, L, mult);
}

/* QuartetCounterLookup.hpp:166-188 */
static void update_quartets(Oracle *o, const Tree *t, int node, const int *etl, size_t L, const int *lteli,
                            uint64_t mult) {
    int cap = t->nodes[node].nchild + 1, k = 0;
    int *ls = (int *)malloc(sizeof(int) * (size_t)cap);
    int act = t->nodes[node].link;
    ls[k++] = act;
    while (ls[0] != t->links[act].next) {
        act = t->links[act].next;
        ls[k++] = act;
    }
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++)
            for (int l = j + 1; l < k; l++) update_three_links(o, t, ls[i], ls[j], ls[l], etl, L, lteli, mult);
    free(ls);
}

static int name_cmp(const void *a, const void *b) {
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

typedef struct { const char *name; int lookup; } NameEnt;
static int nameent_cmp(const void *a, const void *b) {
    return strcmp(((const NameEnt *)a)->name, ((const NameEnt *)b)->name);
}

static size_t cint_bytes(int bits) { return (size_t)bits / 8; }

/*
 * countQuartets (QuartetCounterLookup.hpp:196-238) + ctor (245-273).
 * eval_text: ';'-terminated Newick trees. mult: optional per-tree multiplicity
 * (test harness device for SURVEY Appendix D5: a tree with multiplicity k is
 * exactly k identical consecutive trees). cint_bits 0 = choose by m like
 * QuartetScores.cpp:115-147.
 * Returns 0, or -1 (parse error) / -2 (unknown taxon: the reference lets
 * std::out_of_range escape, QCL:218) / -3 (out of memory).
 */
int qso_count(void *h, const char *eval_text, size_t len, int savemem, int cint_bits, int nthreads,
              const uint64_t *mult, size_t n_mult) {
    Oracle *o = (Oracle *)h;
    const int n = o->n;
    (void)name_cmp;
    /* first pass = countEvalTrees (QuartetScores.cpp:23-32) */
    uint64_t m = 0; size_t ntrees = 0;
    {
        size_t pos = 0;
        while (pos < len) {
            size_t used = 0; char e[256] = {0};
            Tree *t = tree_parse(eval_text + pos, len - pos, &used, e, sizeof e);
            pos += used;
            if (!t) { if (e[0]) { snprintf(o->err, sizeof o->err, "%s", e); return -1; } break; }
            m += (mult && ntrees < n_mult) ? mult[ntrees] : 1;
            ntrees++;
            tree_free(t);
        }
    }
    o->m = m;
    o->savemem = savemem;
    o->bits = cint_bits ? cint_bits : qso_cint_bits_for_m(m);
    o->mask = o->bits == 64 ? ~0ull : ((1ull << o->bits) - 1);
    /* bounded timing runs (qso_set_budget) re-use a table of the same shape: re-allocating and re-touching a
     * 137 GB fast table per thread setting would dominate the leg; its contents are not read afterwards */
    const int reuse = o->budget_s > 0 && o->table_savemem == savemem && o->table_bits == o->bits && (savemem ? o->compact != NULL : o->fast != NULL);
    if (!reuse) { free(o->fast); free(o->compact); o->fast = o->compact = NULL; }
    o->table_savemem = savemem; o->table_bits = o->bits;
    if (reuse) {
    } else if (savemem) {
        o->compact = calloc((size_t)(o->nq * 3 + 1), cint_bytes(o->bits));
        if (!o->compact) { snprintf(o->err, sizeof o->err, "Insufficient memory!"); return -3; }
    } else {
        size_t cells = (size_t)n * (size_t)n * (size_t)n * (size_t)n;
        o->fast = calloc(cells + 1, cint_bytes(o->bits));
        if (!o->fast) { snprintf(o->err, sizeof o->err, "Insufficient memory!"); return -3; }
    }
    /* taxon name -> lookup id */
    NameEnt *ents = (NameEnt *)malloc(sizeof(NameEnt) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { ents[i].name = qso_taxon_name(o, i); ents[i].lookup = i; }
    qsort(ents, (size_t)n, sizeof(NameEnt), nameent_cmp);

#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(savemem ? 1 : nthreads); /* savemem + threads is racy in the reference */
#else
    (void)nthreads;
#endif
    if (o->prefault && !reuse) { /* touch every page of the table outside the timed region (a bounded sample would otherwise time page faults) */
        unsigned char *base = (unsigned char *)(savemem ? o->compact : o->fast);
        const size_t bytes = (savemem ? (size_t)(o->nq * 3 + 1) : (size_t)n * n * n * n + 1) * cint_bytes(o->bits);
#ifdef _OPENMP
        omp_set_num_threads(omp_get_num_procs()); /* first touch with every core, whatever -t the timed run uses */
#endif
#pragma omp parallel for schedule(static)
        for (long long pg = 0; pg < (long long)((bytes + 4095) / 4096); pg++) base[(size_t)pg * 4096] = 0;
#ifdef _OPENMP
        if (nthreads > 0) omp_set_num_threads(savemem ? 1 : nthreads);
#endif
    }
    double t0 = now_s();
    o->budget_t0 = t0; o->budget_hit = 0; o->incr_done = 0;
    size_t pos = 0, ti = 0;
    int rc = 0;
    while (pos < len && rc == 0 && !o->budget_hit) {
        size_t used = 0; char e[256] = {0};
        Tree *t = tree_parse(eval_text + pos, len - pos, &used, e, sizeof e);
        pos += used;
        if (!t) break;
        uint64_t k = (mult && ti < n_mult) ? mult[ti] : 1;
        ti++;
        /* QCL:210-221 */
        int tl; int *seq = eulertour_links(t, &tl);
        int *etl = (int *)malloc(sizeof(int) * (size_t)(t->n_nodes));
        int *lteli = (int *)calloc((size_t)(t->n_links > 0 ? t->n_links : 1), sizeof(int));
        int L = 0;
        for (int i = 0; i < tl && rc == 0; i++) {
            int l = seq[i], x = t->links[l].node;
            if (node_is_leaf(t, x)) {
                NameEnt key; key.name = t->names[x] ? t->names[x] : ""; key.lookup = 0;
                NameEnt *f = (NameEnt *)bsearch(&key, ents, (size_t)n, sizeof(NameEnt), nameent_cmp);
                if (!f) { snprintf(o->err, sizeof o->err, "unknown taxon '%s' in evaluation tree %zu", key.name, ti - 1); rc = -2; break; }
                etl[L++] = f->lookup;
            }
            lteli[l] = L;
        }
        if (rc == 0 && k > 0 && L > 0) {
            const int nEval = t->n_nodes;
            /* QCL:223-228 */
#pragma omp parallel for schedule(dynamic)
            for (int j = 0; j < nEval; j++) {
                if (o->budget_s > 0 && (o->budget_hit || now_s() - o->budget_t0 > o->budget_s)) { o->budget_hit = 1; continue; }
                if (!node_is_leaf(t, j)) update_quartets(o, t, j, etl, (size_t)L, lteli, k);
            }
        }
        free(seq); free(etl); free(lteli);
        tree_free(t);
    }
    o->t_count = now_s() - t0;
    free(ents);
    return rc;
}

static inline uint64_t cell_get(const void *p, int bits, uint64_t i) {
    switch (bits) {
    case 8: return ((const uint8_t *)p)[i];
    case 16: return ((const uint16_t *)p)[i];
    case 32: return ((const uint32_t *)p)[i];
    default: return ((const uint64_t *)p)[i];
    }
}

/* QuartetCounterLookup.hpp:282-290; args are REFERENCE NODE indices; result truncated to CINT */
static uint64_t lookup_quartet_count(const Oracle *o, size_t aIdx, size_t bIdx, size_t cIdx, size_t dIdx) {
    const uint64_t n = (uint64_t)o->n, n2 = n * n, n3 = n2 * n;
    uint64_t a = (uint64_t)o->ref_id_to_lookup[aIdx], b = (uint64_t)o->ref_id_to_lookup[bIdx];
    uint64_t c = (uint64_t)o->ref_id_to_lookup[cIdx], d = (uint64_t)o->ref_id_to_lookup[dIdx];
    uint64_t s = cell_get(o->fast, o->bits, a * n3 + b * n2 + c * n + d) +
                 cell_get(o->fast, o->bits, a * n3 + b * n2 + d * n + c) +
                 cell_get(o->fast, o->bits, b * n3 + a * n2 + c * n + d) +
                 cell_get(o->fast, o->bits, b * n3 + a * n2 + d * n + c);
    return s & o->mask;
}

/* QuartetCounterLookup.hpp:299-318 */
static void count_quartet_occurrences(const Oracle *o, size_t aIdx, size_t bIdx, size_t cIdx, size_t dIdx,
                                      uint64_t *q1, uint64_t *q2, uint64_t *q3) {
    if (o->savemem) {
        uint64_t a = (uint64_t)o->ref_id_to_lookup[aIdx], b = (uint64_t)o->ref_id_to_lookup[bIdx];
        uint64_t c = (uint64_t)o->ref_id_to_lookup[cIdx], d = (uint64_t)o->ref_id_to_lookup[dIdx];
        uint64_t r = qso_rank(a, b, c, d);
        if (r >= o->nq) { /* quartet_lookup_table.hpp:81-83: throw std::runtime_error("id = ..., but quartet_lookup_.size() = ...") */
            Oracle *w = (Oracle *)o;
#pragma omp critical(qso_threw)
            if (!w->threw) {
                snprintf(w->threw_what, sizeof w->threw_what, "id = %llu, but quartet_lookup_.size() = %llu",
                         (unsigned long long)r, (unsigned long long)o->nq);
                w->threw = 1;
            }
            *q1 = *q2 = *q3 = 0;
            return;
        }
        *q1 = cell_get(o->compact, o->bits, r * 3 + (uint64_t)qso_slot(a, b, c, d));
        *q2 = cell_get(o->compact, o->bits, r * 3 + (uint64_t)qso_slot(a, c, b, d));
        *q3 = cell_get(o->compact, o->bits, r * 3 + (uint64_t)qso_slot(a, d, b, c));
    } else {
        *q1 = lookup_quartet_count(o, aIdx, bIdx, cIdx, dIdx);
        *q2 = lookup_quartet_count(o, aIdx, cIdx, bIdx, dIdx);
        *q3 = lookup_quartet_count(o, aIdx, dIdx, bIdx, cIdx);
    }
}

/* Test hook: countQuartetOccurrences on LOOKUP ids. */
int qso_lookup(void *h, int a, int b, int c, int d, uint64_t *out3) {
    Oracle *o = (Oracle *)h;
    if (!o->fast && !o->compact) return -1;
    o->threw = 0; o->threw_what[0] = 0;
    count_quartet_occurrences(o, (size_t)o->lookup_to_node[a], (size_t)o->lookup_to_node[b],
                              (size_t)o->lookup_to_node[c], (size_t)o->lookup_to_node[d], &out3[0], &out3[1], &out3[2]);
    if (o->threw) { snprintf(o->err, sizeof o->err, "%s", o->threw_what); o->threw = 0; return -3; } /* the reference throws */
    return 0;
}

/* Whole table in canonical order: for every a<b<c<d (lookup ids), rank =
 * C(d,4)+C(c,3)+C(b,2)+a, out[rank*3+{0,1,2}] = (ab|cd, ac|bd, ad|bc) as
 * countQuartetOccurrences returns them (savemem: 2x mod 2^bits). */
int qso_get_counts(void *h, uint64_t *out) {
    Oracle *o = (Oracle *)h;
    if (!o->fast && !o->compact) return -1;
    const int n = o->n;
    for (int d = 3; d < n; d++)
        for (int c = 2; c < d; c++)
            for (int b = 1; b < c; b++)
                for (int a = 0; a < b; a++) {
                    uint64_t r = qso_rank((uint64_t)a, (uint64_t)b, (uint64_t)c, (uint64_t)d);
                    qso_lookup(h, a, b, c, d, out + r * 3);
                }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* TreeInformation (TreeInformation.hpp)                                       */
/* ------------------------------------------------------------------------- */

static void tree_information_init(Oracle *o) {
    const Tree *t = o->ref;
    free(o->ti_tour_nodes); free(o->ti_levels); free(o->ti_first); free(o->ti_sparse); free(o->dist_to_root);
    o->dist_to_root = (int *)malloc(sizeof(int) * (size_t)t->n_nodes);
    for (int i = 0; i < t->n_nodes; i++) o->dist_to_root[i] = t->nodes[i].depth; /* node_path_length_vector, TI:96 */
    int len; int *seq = eulertour_links(t, &len);
    o->ti_len = len;
    o->ti_tour_nodes = (int *)malloc(sizeof(int) * (size_t)(len > 0 ? len : 1));
    o->ti_levels = (int *)malloc(sizeof(int) * (size_t)(len > 0 ? len : 1));
    o->ti_first = (int *)malloc(sizeof(int) * (size_t)t->n_nodes);
    for (int i = 0; i < t->n_nodes; i++) o->ti_first[i] = -1;
    for (int i = 0; i < len; i++) { /* TI:103-109 */
        int x = t->links[seq[i]].node;
        o->ti_tour_nodes[i] = x;
        o->ti_levels[i] = o->dist_to_root[x];
        if (o->ti_first[x] < 0) o->ti_first[x] = i;
    }
    free(seq);
    /* RangeMinimumQuery: sparse table of argmin, first minimum on ties */
    int lg = 1; while ((1 << lg) <= (len > 0 ? len : 1)) lg++;
    o->ti_logn = lg;
    o->ti_sparse = (int *)malloc(sizeof(int) * (size_t)lg * (size_t)(len > 0 ? len : 1));
    for (int i = 0; i < len; i++) o->ti_sparse[i] = i;
    for (int j = 1; j < lg; j++)
        for (int i = 0; i + (1 << j) <= len; i++) {
            int x = o->ti_sparse[(size_t)(j - 1) * (size_t)len + (size_t)i];
            int y = o->ti_sparse[(size_t)(j - 1) * (size_t)len + (size_t)(i + (1 << (j - 1)))];
            o->ti_sparse[(size_t)j * (size_t)len + (size_t)i] = (o->ti_levels[y] < o->ti_levels[x]) ? y : x;
        }
}

static inline int rmq_query(const Oracle *o, int i, int j) { /* inclusive [i,j], i<=j */
    int span = j - i + 1, k = 31 - __builtin_clz((unsigned)span);
    int x = o->ti_sparse[(size_t)k * (size_t)o->ti_len + (size_t)i];
    int y = o->ti_sparse[(size_t)k * (size_t)o->ti_len + (size_t)(j - (1 << k) + 1)];
    return (o->ti_levels[y] < o->ti_levels[x]) ? y : x;
}
static inline int rmq_correct_order(const Oracle *o, int i, int j) { /* TI:51-56 */
    return i <= j ? rmq_query(o, i, j) : rmq_query(o, j, i);
}

/* TreeInformation.hpp:71-90 */
static size_t lca_idx(const Oracle *o, size_t u, size_t v, size_t root) {
    int ue = o->ti_first[u], ve = o->ti_first[v], re = o->ti_first[root];
    if ((int)root == o->ref->root) return (size_t)o->ti_tour_nodes[rmq_correct_order(o, ue, ve)];
    size_t c1 = (size_t)o->ti_tour_nodes[rmq_correct_order(o, ue, ve)];
    size_t c2 = (size_t)o->ti_tour_nodes[rmq_correct_order(o, ue, re)];
    size_t c3 = (size_t)o->ti_tour_nodes[rmq_correct_order(o, ve, re)];
    if (c1 == c2) return c3;
    else if (c1 == c3) return c2;
    else return c1;
}

/* TreeInformation.hpp:40-43 */
static unsigned distance_in_edges(const Oracle *o, size_t u, size_t v) {
    size_t l = lca_idx(o, u, v, (size_t)o->ref->root);
    return (unsigned)(o->dist_to_root[u] + o->dist_to_root[v] - 2 * o->dist_to_root[l]);
}

/* ------------------------------------------------------------------------- */
/* Scoring (QuartetScoreComputer.hpp)                                          */
/* ------------------------------------------------------------------------- */

/* QuartetScoreComputer.hpp:135-159 */
double qso_log_score(uint64_t q1, uint64_t q2, uint64_t q3) {
    if (q1 == 0 && q2 == 0 && q3 == 0) return 0;
    uint64_t sum = q1 + q2 + q3;
    double p_q1 = (double)q1 / sum;
    double p_q2 = (double)q2 / sum;
    double p_q3 = (double)q3 / sum;
    double qic = 1;
    if (p_q1 != 0) qic += p_q1 * log(p_q1) / log(3);
    if (p_q2 != 0) qic += p_q2 * log(p_q2) / log(3);
    if (p_q3 != 0) qic += p_q3 * log(p_q3) / log(3);
    if (q1 < q2 || q1 < q3) return qic * -1;
    else return qic;
}

/* path_set(start, finish, lca): start -> ... -> lca (flagged), then finish -> ...
 * -> child of lca. Each item is a node; .link() is its primary link, .edge()
 * that link's edge. Calls f(node, is_lca). */
typedef void (*path_fn)(const Oracle *o, int node, int is_lca, void *ctx);
static void path_set(const Oracle *o, int start, int finish, int lca, path_fn f, void *ctx) {
    const Tree *t = o->ref;
    int x = start;
    while (x != lca) { f(o, x, 0, ctx); x = t->nodes[x].parent; }
    f(o, lca, 1, ctx);
    x = finish;
    while (x != lca) { f(o, x, 0, ctx); x = t->nodes[x].parent; }
}

static void last_link_fn(const Oracle *o, int node, int is_lca, void *ctx) {
    if (is_lca) return;
    *(int *)ctx = o->ref->links[node_primary_link(o->ref, node)].outer;
}

/* QuartetScoreComputer.hpp:169-194 */
static int get_path_inner_links(const Oracle *o, int u, int v, int lca, int *ul, int *vl) {
    if (u == v) return -1; /* reference throws */
    if (u == lca) {
        int l = -1;
        path_set(o, v, u, lca, last_link_fn, &l);
        *ul = l; *vl = node_primary_link(o->ref, v);
        return 0;
    }
    if (v == lca) {
        int l = -1;
        path_set(o, u, v, lca, last_link_fn, &l);
        *ul = node_primary_link(o->ref, u); *vl = l;
        return 0;
    }
    *ul = node_primary_link(o->ref, u); *vl = node_primary_link(o->ref, v);
    return 0;
}

typedef struct { double *arr; double val; } MinCtx;
static void min_edge_fn(const Oracle *o, int node, int is_lca, void *ctx) {
    if (is_lca) return;
    MinCtx *m = (MinCtx *)ctx;
    int e = o->ref->links[node_primary_link(o->ref, node)].edge;
#pragma omp critical(qso_minupd)
    { m->arr[e] = (m->val < m->arr[e]) ? m->val : m->arr[e]; } /* std::min(old,new) */
}

/* QSC:436-454: from/to re-derivation and LQ update for one quartet with ref topology ab|cd */
static void lq_update(Oracle *o, size_t aIdx, size_t bIdx, size_t cIdx, size_t dIdx, double qic) {
    size_t root = (size_t)o->ref->root;
    size_t lca_ab = lca_idx(o, aIdx, bIdx, root);
    size_t lca_cd = lca_idx(o, cIdx, dIdx, root);
    size_t from, to;
    if (lca_cd == lca_idx(o, cIdx, dIdx, lca_ab)) {
        from = lca_idx(o, aIdx, bIdx, lca_cd);
        to = lca_cd;
    } else {
        from = lca_ab;
        to = lca_idx(o, cIdx, dIdx, lca_ab);
    }
    size_t l = lca_idx(o, from, to, root);
    MinCtx mc = { o->LQ, qic };
    path_set(o, (int)from, (int)to, (int)l, min_edge_fn, &mc);
}

/* QuartetScoreComputer.hpp:379-490 */
static void process_node_pair(Oracle *o, size_t uIdx, size_t vIdx) {
    const Tree *t = o->ref;
    unsigned p1 = 0, p2 = 0, p3 = 0;       /* 32-bit, wraps (QSC:382) */
    uint64_t P1 = 0, P2 = 0, P3 = 0;       /* 64-bit alternative (qp_exact64) */
    size_t lcaIdx = lca_idx(o, uIdx, vIdx, (size_t)t->root);
    int il1, il2;
    if (get_path_inner_links(o, (int)uIdx, (int)vIdx, (int)lcaIdx, &il1, &il2) != 0) return;
    int ls1 = t->links[il1].next, ls2 = t->links[t->links[il1].next].next;
    int ls3 = t->links[il2].next, ls4 = t->links[t->links[il2].next].next;
    const size_t L = (size_t)o->n_etl;
    const int *lt = o->linkToEulerLeafIndex;
    size_t s1 = (size_t)lt[ls1] % L, e1 = (size_t)lt[t->links[ls1].outer] % L;
    size_t s2 = (size_t)lt[ls2] % L, e2 = (size_t)lt[t->links[ls2].outer] % L;
    size_t s3 = (size_t)lt[ls3] % L, e3 = (size_t)lt[t->links[ls3].outer] % L;
    size_t s4 = (size_t)lt[ls4] % L, e4 = (size_t)lt[t->links[ls4].outer] % L;
    size_t ai = s1, bi = s2, ci = s3, di = s4;
    while (ai != e1) {
        while (bi != e2) {
            while (ci != e3) {
                while (di != e4) {
                    size_t a = o->eulerTourLeaves[ai], b = o->eulerTourLeaves[bi];
                    size_t c = o->eulerTourLeaves[ci], d = o->eulerTourLeaves[di];
                    uint64_t q1, q2, q3;
                    count_quartet_occurrences(o, a, b, c, d, &q1, &q2, &q3);
                    if (o->threw) return; /* the exception leaves processNodePair (and ends the reference's run) */
                    p1 += (unsigned)q1; p2 += (unsigned)q2; p3 += (unsigned)q3;
                    P1 += q1; P2 += q2; P3 += q3;
                    double qic = qso_log_score(q1, q2, q3);
                    lq_update(o, a, b, c, d, qic);
                    di = (di + 1) % L;
                }
                ci = (ci + 1) % L;
                di = s4;
            }
            bi = (bi + 1) % L;
            ci = s3; di = s4;
        }
        ai = (ai + 1) % L;
        bi = s2; ci = s3; di = s4;
    }
    double qpic = o->qp_exact64 ? qso_log_score(P1, P2, P3) : qso_log_score(p1, p2, p3);
    int u_link = t->nodes[uIdx].link, v_link = t->nodes[vIdx].link;
    if (t->links[t->links[u_link].outer].node == (int)vIdx) o->QP[t->links[u_link].edge] = qpic;
    else if (t->links[t->links[v_link].outer].node == (int)uIdx) o->QP[t->links[v_link].edge] = qpic;
    MinCtx mc = { o->EQP, qpic };
    path_set(o, (int)uIdx, (int)vIdx, (int)lcaIdx, min_edge_fn, &mc);
}

/* QuartetScoreComputer.hpp:513-593 (and the same topology test in :623-690) */
static int ref_topology(Oracle *o, size_t u, size_t v, size_t w, size_t z, size_t *a, size_t *b, size_t *c, size_t *d) {
    size_t root = (size_t)o->ref->root;
    size_t lca_uv = lca_idx(o, u, v, root), lca_uw = lca_idx(o, u, w, root), lca_uz = lca_idx(o, u, z, root);
    size_t lca_vw = lca_idx(o, v, w, root), lca_vz = lca_idx(o, v, z, root), lca_wz = lca_idx(o, w, z, root);
    unsigned d1 = distance_in_edges(o, lca_uv, lca_wz), d2 = distance_in_edges(o, lca_uw, lca_vz),
             d3 = distance_in_edges(o, lca_uz, lca_vw);
    if (d1 > d2 && d1 > d3) { *a = u; *b = v; *c = w; *d = z; return 1; }
    if (d2 > d1 && d2 > d3) { *a = u; *b = w; *c = v; *d = z; return 1; }
    if (d3 > d1 && d3 > d2) { *a = u; *b = z; *c = v; *d = w; return 1; }
    return 0;
}

static void scores_multifurcating(Oracle *o) {
    const int L = o->n_etl;
#pragma omp parallel for schedule(dynamic)
    for (int ui = 0; ui < L; ui++)
        for (int vi = ui + 1; vi < L; vi++)
            for (int wi = vi + 1; wi < L; wi++)
                for (int zi = wi + 1; zi < L; zi++) {
                    size_t a, b, c, d;
                    if (!ref_topology(o, o->eulerTourLeaves[ui], o->eulerTourLeaves[vi], o->eulerTourLeaves[wi],
                                      o->eulerTourLeaves[zi], &a, &b, &c, &d))
                        continue;
                    uint64_t q1, q2, q3;
                    count_quartet_occurrences(o, a, b, c, d, &q1, &q2, &q3);
                    lq_update(o, a, b, c, d, qso_log_score(q1, q2, q3));
                }
}

/* is_bifurcating(tree): max over nodes of (degree-1) == 2 (QSC:760).
 * A degree-2 root therefore passes as "bifurcating" (SURVEY quirk Q5). */
static int tree_is_bifurcating(const Tree *t) {
    int max_rank = 0;
    for (int i = 0; i < t->n_nodes; i++) {
        int deg = t->nodes[i].nchild + (t->nodes[i].parent >= 0 ? 1 : 0);
        if (deg - 1 > max_rank) max_rank = deg - 1;
    }
    return max_rank == 2;
}

int qso_is_bifurcating(void *h) { return tree_is_bifurcating(((Oracle *)h)->ref); }

/* ctor tail, QuartetScoreComputer.hpp:760-778 */
int qso_score(void *h, int nthreads, int qp_exact64) {
    Oracle *o = (Oracle *)h;
    if (!o->fast && !o->compact) { snprintf(o->err, sizeof o->err, "count first"); return -1; }
    const Tree *t = o->ref;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    o->qp_exact64 = qp_exact64;
    o->threw = 0; o->threw_what[0] = 0;
    tree_information_init(o);
    free(o->LQ); free(o->QP); free(o->EQP);
    size_t ne = (size_t)(t->n_edges > 0 ? t->n_edges : 1);
    o->LQ = (double *)malloc(sizeof(double) * ne);
    o->QP = (double *)malloc(sizeof(double) * ne);
    o->EQP = (double *)malloc(sizeof(double) * ne);
    for (size_t i = 0; i < ne; i++) o->LQ[i] = o->QP[i] = o->EQP[i] = INFINITY;
    double t0 = now_s();
    if (!tree_is_bifurcating(t)) {
        o->have_qp = 0;
        scores_multifurcating(o);
    } else {
        o->have_qp = 1;
        /* QSC:495-508 */
#pragma omp parallel for schedule(dynamic)
        for (int i = 0; i < t->n_nodes; i++) {
            if (node_is_leaf(t, i)) continue;
            for (int j = i + 1; j < t->n_nodes; j++) {
                if (node_is_leaf(t, j) || o->threw) continue;
                process_node_pair(o, (size_t)i, (size_t)j);
            }
        }
    }
    o->t_score = now_s() - t0;
    if (o->threw) { /* the reference terminates with this what() (uncaught std::runtime_error) */
        snprintf(o->err, sizeof o->err, "%s", o->threw_what);
        return -3;
    }
    return 0;
}

/* Scores by edge index; qp/eqp are empty (return 1) for a multifurcating reference. */
int qso_get_scores(void *h, double *lq, double *qp, double *eqp) {
    Oracle *o = (Oracle *)h;
    if (!o->LQ) return -1;
    size_t ne = (size_t)o->ref->n_edges;
    memcpy(lq, o->LQ, sizeof(double) * ne);
    if (o->have_qp) {
        if (qp) memcpy(qp, o->QP, sizeof(double) * ne);
        if (eqp) memcpy(eqp, o->EQP, sizeof(double) * ne);
        return 0;
    }
    return 1;
}

/* member[lookup id] = 1 if the taxon is on the child (secondary) side of the edge */
int qso_edge_side(void *h, int edge, uint8_t *member) {
    Oracle *o = (Oracle *)h;
    const Tree *t = o->ref;
    int child = t->links[t->edge_secondary[edge]].node;
    memset(member, 0, (size_t)o->n);
    for (int i = 0; i < o->n; i++) {
        int x = o->lookup_to_node[i];
        while (x >= 0 && x != child) x = t->nodes[x].parent;
        if (x == child) member[i] = 1;
    }
    return 0;
}

/* QuartetScoreComputer.hpp:623-690: one line per resolved-in-reference quartet,
 * "(a,b|c,d): qic\n" with operator<<(double) default formatting (%g, 6 digits). */
int qso_raw_qic(void *h, const char *path) {
    Oracle *o = (Oracle *)h;
    if (!o->ti_first) tree_information_init(o);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    const int L = o->n_etl;
    for (int ui = 0; ui < L; ui++)
        for (int vi = ui + 1; vi < L; vi++)
            for (int wi = vi + 1; wi < L; wi++)
                for (int zi = wi + 1; zi < L; zi++) {
                    size_t a, b, c, d;
                    if (!ref_topology(o, o->eulerTourLeaves[ui], o->eulerTourLeaves[vi], o->eulerTourLeaves[wi],
                                      o->eulerTourLeaves[zi], &a, &b, &c, &d))
                        continue;
                    uint64_t q1, q2, q3;
                    count_quartet_occurrences(o, a, b, c, d, &q1, &q2, &q3);
                    double qic = qso_log_score(q1, q2, q3);
                    fprintf(f, "(%s,%s|%s,%s): %g\n", o->ref->names[a] ? o->ref->names[a] : "",
                            o->ref->names[b] ? o->ref->names[b] : "", o->ref->names[c] ? o->ref->names[c] : "",
                            o->ref->names[d] ? o->ref->names[d] : "", qic);
                }
    fclose(f);
    return 0;
}

#ifdef QSO_MAIN
/* tiny driver: qs_oracle ref.nwk eval.nwk [savemem] [threads] -> prints timings */
static char *slurp(const char *p, size_t *len) {
    FILE *f = fopen(p, "rb"); if (!f) return NULL;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)n + 1); if (fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return NULL; }
    b[n] = 0; fclose(f); *len = (size_t)n; return b;
}
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s ref.nwk eval.nwk [savemem=0] [threads=1]\n", argv[0]); return 2; }
    size_t rl, el; char *r = slurp(argv[1], &rl), *e = slurp(argv[2], &el);
    if (!r || !e) { fprintf(stderr, "cannot read input\n"); return 1; }
    char err[256] = {0};
    void *h = qso_create(r, err, sizeof err);
    if (!h) { fprintf(stderr, "%s\n", err); return 1; }
    int sm = argc > 3 ? atoi(argv[3]) : 0, th = argc > 4 ? atoi(argv[4]) : 1;
    if (qso_count(h, e, el, sm, 0, th, NULL, 0) != 0) { fprintf(stderr, "%s\n", qso_last_error(h)); return 1; }
    qso_score(h, th, 0);
    printf("n=%d count_s=%.6f score_s=%.6f\n", qso_n_taxa(h), qso_time_count(h), qso_time_score(h));
    qso_destroy(h);
    return 0;
}
#endif
