/*
 * gnx_hip.h - C-ABI of libgnxhip.so: the MI355X (gfx950) implementation of
 * Geonomics' per-generation simulation loop.
 *
 * The reference (erthward/geonomics 1.4.9) is pure Python and has no FFI; the
 * boundary it offers is the Python object API.  Each entry point below names
 * the reference method whose work it replaces (paths relative to
 * geonomics/ in the reference).  The Python host layer in geonomics_amd/
 * binds these with ctypes (geonomics_amd/_native.py); INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on failure;
 *     gnx_last_error() then returns a message.
 *   - the library owns all device memory behind the opaque handle; host
 *     buffers passed in are copied; downloads write to caller-allocated buffers.
 *   - one host thread per handle; calls are ordered on the handle's HIP stream
 *     and synchronous at download / count calls.
 *   - rasters are float32 [H][W] (row = y, col = x), the reference's
 *     Layer.rast[y, x] order (structs/landscape.py Layer; dim = (x, y)).
 *   - genotypes are bit-packed: per individual 2 homologues x W64 u64 words,
 *     bit l of homologue h == Individual.g[l, h] (structs/individual.py:103).
 *     W64 = gnx_words_per_hom(L).
 *   - extinction is not an error: N == 0 after a step (structs/species.py:841).
 */
#ifndef GNX_HIP_H
#define GNX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gnx_state gnx_state;

/* ---- enums -------------------------------------------------------------- */
enum { GNX_DIST_LOGNORMAL = 0, GNX_DIST_WALD = 1, GNX_DIST_LEVY = 2 };
enum { GNX_MATE_UNIFORM = 0, GNX_MATE_NEAREST = 1, GNX_MATE_INVERSE = 2 };
enum { GNX_SURF_NONE = 0, GNX_SURF_MIXTURE = 1, GNX_SURF_UNIMODAL = 2 };

/* fields for gnx_download() */
enum {
  GNX_F_X = 0, GNX_F_Y = 1, GNX_F_AGE = 2, GNX_F_SEX = 3, GNX_F_ID = 4,
  GNX_F_E = 5,      /* float [n_layers][N]   */
  GNX_F_Z = 6,      /* float [n_traits][N]   */
  GNX_F_FIT = 7, GNX_F_GROW = 8,
  GNX_F_GENO = 9    /* uint64 [N][2][W64], in slot order */
};
/* rasters for gnx_download_raster(): double [H][W] */
enum { GNX_R_N = 0, GNX_R_NPAIRS = 1, GNX_R_K = 2, GNX_R_D = 3, GNX_R_COUNTS = 4 };

/* kernels for gnx_kernel_time() */
enum {
  GNX_K_MOVE = 0, GNX_K_SORT = 1, GNX_K_PERMUTE = 2, GNX_K_FIND_MATES = 3,
  GNX_K_PAIRS = 4, GNX_K_OFFSPRING = 5, GNX_K_CROSSOVER = 6,
  GNX_K_PHENOTYPE = 7, GNX_K_DENSITY = 8, GNX_K_DEATH = 9, GNX_K_COMPACT = 10,
  GNX_K_CROSSOVER_TAIL = 11,   /* the narrow share of a split crossover launch */
  GNX_K_COUNT = 12
};

/* ---- configuration ------------------------------------------------------ */
typedef struct {
  int32_t W, H;            /* landscape dim (x, y): Landscape.dim            */
  int32_t n_layers;
  int32_t L;               /* loci; 0 = species without gen_arch             */
  int32_t n_traits;
  int64_t cap_inds;        /* individual-slot capacity                       */
  int64_t cap_rows;        /* genome-row capacity (0 if L == 0)              */
  uint64_t seed;           /* params.model.seed.num (sim/model.py:98-99)     */
  int32_t device;          /* HIP device ordinal                             */
  int32_t reserved;
} gnx_config;

/* Species life-history parameters: the 'mating', 'mortality' and 'movement'
 * sections of the parameters file (sim/params.py SPP_PARAMS), hoisted to
 * Species attributes at structs/species.py:409-425.                          */
typedef struct {
  /* mating */
  double  b;                       /* P(pair mates)                          */
  double  R;                       /* intrinsic growth rate                  */
  double  n_births_lambda;
  int32_t n_births_fixed;
  int32_t sexed;                   /* mating.sex                             */
  double  p_male;                  /* sex_ratio/(sex_ratio+1) (species.py:416) */
  double  mating_radius;           /* < 0 => None (panmixia)                 */
  int32_t mate_mode;               /* GNX_MATE_*                             */
  int32_t repro_age[2];            /* [female, male]; both equal if unsexed  */
  /* mortality */
  int32_t max_age;                 /* < 0 => None                            */
  double  d_min, d_max;
  double  window_width;            /* density_grid_window_width; <=0 => None */
  /* movement */
  int32_t move;
  double  dir_mu, dir_kappa;
  int32_t move_distr;              /* GNX_DIST_*                             */
  double  move_p1, move_p2;
  int32_t disp_distr;
  double  disp_p1, disp_p2;
  int32_t move_surf;               /* GNX_SURF_*                             */
  int32_t move_surf_layer;
  double  move_surf_kappa;
  int32_t disp_surf;
  int32_t disp_surf_layer;
  double  disp_surf_kappa;
  double  res_ratio[2];            /* Landscape._res_ratio                   */
  /* carrying capacity: K = rast[K_layer] * K_factor (species.py:546)        */
  int32_t K_layer;
  int32_t pad0;
  double  K_factor;
} gnx_species_params;

/* ---- lifecycle ---------------------------------------------------------- */
int  gnx_create(const gnx_config* cfg, gnx_state** out);
void gnx_destroy(gnx_state* h);
const char* gnx_last_error(void);
int  gnx_words_per_hom(int32_t L);
/* blocks a homologue is stored in (a block = W64*8 / n bytes is the unit the crossover copies
 * or shares with the parent, csrc/gnx_half.h); 0 without genomes                              */
int  gnx_blocks_per_hom(const gnx_state* h);
/* use an externally created hipStream_t (e.g. torch's current stream)       */
int  gnx_set_stream(gnx_state* h, void* hip_stream);
int  gnx_synchronize(gnx_state* h);

/* ---- landscape / species setup ----------------------------------------- */
/* Landscape layers: lyr.rast for every Layer (structs/landscape.py:34-120)  */
int gnx_upload_rasters(gnx_state* h, const float* rasts /*[n_layers][H][W]*/);
int gnx_upload_layer(gnx_state* h, int32_t layer, const float* rast);
/* Species.K given explicitly, double [H][W] (demographic change events scale it:
 * ops/change.py:633-651); NULL returns to rast[K_layer] * K_factor            */
int gnx_set_k_raster(gnx_state* h, const double* K);
int gnx_set_species_params(gnx_state* h, const gnx_species_params* p);

/* Population: _make_species / _make_individual (structs/species.py:3300-3320,
 * structs/individual.py:188-228).  ids must be ascending.                   */
int gnx_upload_population(gnx_state* h, int64_t N, const float* x,
                          const float* y, const int32_t* age,
                          const uint8_t* sex, const int64_t* id);
/* N individuals at uniform random positions, ids 0..N-1, age 0              */
int gnx_init_population(gnx_state* h, int64_t N);

/* ---- genomic architecture ----------------------------------------------- */
/* Recombinations._subsetters (structs/genome.py:188-230) as path bits:
 * paths[k][w] bit l = homologue the k-th cached path is on at locus l.      */
int gnx_set_recomb_paths(gnx_state* h, int32_t n, const uint64_t* paths);
/* Trait (structs/genome.py:284-438): loci ascending, alpha per locus.
 * phi_rast == NULL => scalar phi.                                           */
int gnx_set_trait(gnx_state* h, int32_t t, int32_t n_loci, const int32_t* loci,
                  const double* alpha, int32_t layer, double phi,
                  const float* phi_rast, double gamma, int32_t univ_adv);
/* GenomicArchitecture.dom (structs/genome.py:552-555); NULL => codominant   */
int gnx_set_dominance(gnx_state* h, const uint8_t* dom /*[L]*/);
/* delet_loci / delet_loci_s (structs/genome.py:589-591)                     */
int gnx_set_deleterious(gnx_state* h, int32_t n, const int32_t* loci,
                        const double* s);
/* Individuals' genomes, in current slot order (inject / restore)            */
int gnx_upload_genomes(gnx_state* h, const uint64_t* geno /*[N][2][W64]*/);
/* Species._set_genomes_and_tables + _make_starting_mutations
 * (structs/species.py:956-967,1087; structs/genome.py:1108-1157):
 * exactly n_per_site[l] of the 2N homologues carry a 1 at site l.  The k-th
 * genome drawn goes to the individual with the k-th smallest id (the order the
 * reference walks its individuals in), whatever order the slots are in.     */
int gnx_assign_genomes(gnx_state* h, const int32_t* n_per_site /*[L]*/);
/* recompute all phenotypes (Species._set_z, structs/species.py:925)         */
int gnx_set_z(gnx_state* h);
/* phenotypes of slots [first, first+n) only (Species._set_z_individ, :929)   */
int gnx_set_z_range(gnx_state* h, int64_t first, int64_t n);

/* ---- one time step ------------------------------------------------------- */
/* Species._set_age_stage (structs/species.py:567)                           */
int gnx_age(gnx_state* h);
/* Species._do_movement (structs/species.py:582-585; ops/movement.py:34-95)  */
int gnx_move(gnx_state* h);
/* Species._do_pop_dynamics (structs/species.py:822; ops/demography.py:183):
 * pairs -> n_pairs density -> mating (births, dispersal, crossover,
 * phenotype) -> N density -> d -> death probabilities -> mortality.
 * burn != 0: no genomes/selection (burn-in).  Appends to Nt/births/deaths
 * counters readable with gnx_counts().                                      */
int gnx_pop_dynamics(gnx_state* h, int32_t burn, int32_t with_selection);
/* the same in two halves, so that the host can apply this step's mutations to
 * the new offspring in between, where the reference does
 * (structs/species.py:807-809): _mate = pairs, n_pairs density, births;
 * _die = N density, d, death probabilities, mortality.  After _mate the
 * offspring occupy slots [N_before, N_before + births).                     */
int gnx_pop_dynamics_mate(gnx_state* h, int32_t burn);
int gnx_pop_dynamics_die(gnx_state* h, int32_t burn, int32_t with_selection);
/* whole fn-queue entry for one step: age, move (if params.move), pop dynamics
 * (sim/model.py:603-667)                                                    */
int gnx_step(gnx_state* h, int32_t burn, int32_t with_selection);
/* gnx_step in three parts - _begin: age, movement, cell sort, mate search, pair list
 * enqueued; _mid: pair count read, births, densities, death probabilities, death draws,
 * compaction and crossover enqueued; _end: survivor counts read - and gnx_step_many: one
 * step of n INDEPENDENT handles (the iterations of a model, sim/model.py:866-953, whose
 * TODO at :924-925 wants them farmed out), all first parts, then all second, then all
 * third, so that the handles' kernels run side by side on their own streams while one
 * host thread drives them.  Same results as gnx_step on each handle.                     */
int gnx_step_begin(gnx_state* h, int32_t burn);
int gnx_step_mid(gnx_state* h, int32_t burn, int32_t with_selection);
int gnx_step_end(gnx_state* h, int32_t burn);
int gnx_step_many(gnx_state** hs, int32_t n, int32_t burn, int32_t with_selection);
/* T time steps without the host in the loop - Model.walk(T) of the reference
 * (sim/model.py:966-1161: T times _do_timestep over the function queue, :699-744).
 * gnx_step reads the pair count and the survivor count back in every step and sizes the
 * next kernels with them; gnx_walk keeps every count of the step in device memory, sizes
 * the grids by the handle's capacity and replays one captured HIP graph per step: one
 * runtime call and no read-back per step (at 10^5 individuals - BASELINE configs[1], [2] -
 * the host-driven step is bound by the ~45 runtime calls it makes, not by its kernels).
 * Same draws, same canonical orders, same kernels: the population equals the one T calls of
 * gnx_step leave, id by id.  A handle the device-driven step does not cover (tiles,
 * panmixia, Poisson births, no movement, profiling on; GNX_DD=0) walks through gnx_step.
 * The population must fit the capacity throughout: a step whose offspring do not fit is
 * reported as an error when the walk ends.
 * gnx_walk_many: the same for n INDEPENDENT handles (the iterations of one model,
 * sim/model.py:866-953, TODO at :924-925), step t of every handle enqueued before step
 * t + 1 of any, so that their kernels share the chip.
 * gnx_walk_history: (N at the start, births, deaths) of the last max_steps steps of the
 * last walk - Species.Nt / n_births / n_deaths (structs/species.py:374-380, 554) - returns
 * how many were written.                                                                  */
int gnx_walk(gnx_state* h, int64_t T, int32_t burn, int32_t with_selection);
int gnx_walk_many(gnx_state** hs, int32_t n, int64_t T, int32_t burn, int32_t with_selection);
int64_t gnx_walk_history(gnx_state* h, int64_t max_steps, int64_t* n_start, int64_t* births,
                         int64_t* deaths);
int gnx_counts(gnx_state* h, int64_t* N, int64_t* births, int64_t* deaths);
/* Running totals over the gnx_step calls since the last gnx_reset_totals (what Species.Nt /
 * n_births / n_deaths accumulate step by step, structs/species.py:554,  kept in the library
 * so that a driver loop need not call back between steps): out[6] = steps, sum of N at the
 * START of each step (the metric's individual-timesteps), births, deaths, births whose
 * genomes the crossover wrote, steps that gnx_walk took the device-driven way.  Host-side
 * bookkeeping only: no device access.                                                       */
int gnx_totals(gnx_state* h, int64_t* out);
int gnx_reset_totals(gnx_state* h);
/* Where the new offspring's genomes are cut (ops/mating.py:130-214, the crossover).
 * on (default, one GPU): after the step's death draws, for the offspring that survive
 * them only, on a second HIP stream under the next step's kernels - offspring that die at
 * age 0 (structs/species.py:822-833 kills them in the same _do_pop_dynamics call) never
 * get a genome.  off: for every birth, inside gnx_pop_dynamics_mate.  Same results either
 * way (draws are keyed by id); any genome access in between triggers the off path.      */
int gnx_set_defer_crossover(gnx_state* h, int32_t on);
/* How the deferred crossover shares the GPU with the next step's kernels.  0 (default):
 * it runs at full width beside the compaction, the sort index's compaction and the next
 * movement; the next cell sort waits for it.  1: a narrow crossover runs beside the
 * WHOLE next step.  2: nothing runs beside it (the kernel's own rate; a slower step).
 * Results do not depend on the mode.                                                   */
int gnx_set_crossover_overlap(gnx_state* h, int32_t mode);
/* Split every deferred crossover launch: wide_per_1024 / 1024 of its jobs at full width
 * (the next cell sort waits for them), the rest as a narrow launch that shares the chip
 * with the sort and the kernels after it.  0 or 1024 = one launch.  Results do not depend
 * on it. */
int gnx_set_crossover_split(gnx_state* h, int32_t wide_per_1024);
/* Bookkeeping of the shared genome blocks (where a gamete's path has no switch point the
 * child refers to the parent's block instead of copying it; blocks nobody alive refers to
 * are found by a mark-and-sweep collection when the free stack runs low).  Runs a
 * collection, then out[6] = logical blocks of the individuals that have a genome row / 2,
 * broken references, collections so far, physical blocks in use, free physical blocks,
 * physical blocks in all.  Consistent iff out[1] == 0 and out[3] + out[4] == out[5]. */
int gnx_debug_halves(gnx_state* h, int64_t* out);
/* The same bookkeeping as the host sees it, WITHOUT touching the device (no join of a
 * crossover in flight, no collection): out[8] = blocks per homologue, words per block,
 * collections so far, row spread, sparse paths (0 / 1), free blocks the host counts on,
 * free logical rows, 1 while offspring still wait for their deferred crossover.        */
int gnx_genome_info(gnx_state* h, int64_t* out);
/* measurement: the job list of the last crossover, 16 bytes per copied block {the
 * parent's two physical blocks, the block written, (path * 2 + start homologue) | block
 * index << 24} (csrc/gnx_xo.h); blocks without a switch point are not in it            */
int gnx_last_crossover_jobs(gnx_state* h, void* dst, int64_t max_jobs, int64_t* n_jobs);
/* births whose genomes the last crossover wrote (== births when not deferred)          */
int64_t gnx_last_crossover_births(gnx_state* h);
int64_t gnx_step_index(gnx_state* h);
int gnx_set_step_index(gnx_state* h, int64_t step);

/* mutation (ops/mutation.py:62-131): set bit (locus, hom) of listed slots   */
int gnx_mutate(gnx_state* h, int32_t n, const int64_t* slot,
               const int32_t* locus, const uint8_t* hom);

/* ---- read-back ------------------------------------------------------------ */
/* slot order == ascending id order is NOT guaranteed; download GNX_F_ID and
 * sort on the host (geonomics_amd does).                                    */
int gnx_download(gnx_state* h, int32_t field, void* dst, int64_t dst_bytes);
/* genotypes of selected slots: uint64 [n][2][W64]                           */
/* new coordinates of all N individuals, slot order (Individual.x / .y assigned by a
 * script + Species._set_coords_and_cells, structs/species.py:937-939); e follows */
int gnx_set_positions(gnx_state* h, const float* x /*[N]*/, const float* y /*[N]*/);
/* slots in use = rows of a gnx_download (the tile's ghosts included while they are
 * resident, i.e. between gnx_tile_import_ghosts and gnx_tile_die)              */
int64_t gnx_n_slots(gnx_state* h);
int gnx_download_genomes(gnx_state* h, int64_t n, const int64_t* slots,
                         uint64_t* dst);
/* double [H][W]: N (Species.N), n_pairs, K, d as of the last pop_dynamics;
 * GNX_R_COUNTS = individuals per cell (sim/burnin.py:44-59)                 */
int gnx_download_raster(gnx_state* h, int32_t which, double* dst);
/* burn-in spatial tester (sim/burnin.py:44-59): updates the per-cell count
 * raster and returns mean and std of (counts_now - counts_prev)             */
int gnx_spatial_diff_stats(gnx_state* h, double* mean, double* std);
/* the same update, returning the sums behind them (integers: sum of the per-cell count
 * differences and of their squares), which add exactly over the tiles of a tiled run  */
int gnx_spatial_diff_sums(gnx_state* h, double* sum, double* sum_sq);

/* ---- operator-level entry points (parity tests; explicit random inputs) -- */
/* ops/movement.py:74-92 with injected direction/distance draws              */
int gnx_op_move(gnx_state* h, const float* theta, const float* dist);
/* draws only: what gnx_move would draw for the current population           */
int gnx_op_move_draws(gnx_state* h, float* theta, float* dist);
/* structs/species.py:2157-2215 + ops/mating.py:24-117.  Outputs mate slot
 * per individual (-1 none) and the final pair list; keep == NULL => draw
 * Bernoulli(b) from the device stream.                                      */
int gnx_op_find_pairs(gnx_state* h, const uint8_t* keep, int32_t* mate,
                      int32_t* pairs /*[cap][2]*/, int64_t* n_pairs);
/* ops/mating.py:130-214: B offspring from parent slots, path keys and start
 * homologues; children are appended to the population (rows allocated).     */
int gnx_op_crossover(gnx_state* h, int64_t B, const int32_t* parent_slots,
                     const int32_t* keys, const uint8_t* start_homs);
/* ops/movement.py:98-141: offspring positions from A attempts of draws      */
int gnx_op_dispersal(gnx_state* h, int64_t B, int32_t A, const float* mid_x,
                     const float* mid_y, const float* theta,
                     const float* dist, float* out_x, float* out_y,
                     int32_t* attempt_used);
/* utils/spatial.py:73-146: density raster of arbitrary points               */
int gnx_op_density(gnx_state* h, int64_t n, const float* x, const float* y,
                   double* node_vals /*[Jy][Jx] or NULL*/, double* raster);
int gnx_density_lattice_dims(gnx_state* h, int32_t* Jx, int32_t* Jy);
/* ops/demography.py:116: N.max(), the maximum of the individuals' density raster
 * the last death probabilities (gnx_step, gnx_pop_dynamics_*, gnx_op_death_probs)
 * clipped dNdt with; waits for the handle's stream                            */
int gnx_density_nmax(gnx_state* h, double* nmax);
/* ops/demography.py:253-321 + ops/selection.py:119-125: death probabilities
 * of the current population given node densities for N and n_pairs          */
int gnx_op_death_probs(gnx_state* h, int32_t with_selection,
                       const double* nodes_N, const double* nodes_pairs,
                       double* p_death, double* d_at_cell);
/* ops/demography.py:175-180 with an injected death mask (by slot); the survivors
 * keep their order.  (The mortality of gnx_step / gnx_pop_dynamics_die leaves the
 * survivors of the first N - deaths slots where they are and moves the others
 * into the slots of the dead: slot order carries no meaning between steps.) */
int gnx_op_mortality(gnx_state* h, const uint8_t* dead);

/* ---- spatial tiling over several GPUs ---------------------------
 * The reference has no distributed mode; these entry points are new.  One
 * process/GPU owns one tile of a uniform R x C grid and the individuals inside
 * it; the host layer (geonomics_amd/parallel.py) moves the staged buffers with
 * torch.distributed (RCCL).  Step order on a tiled landscape:
 *   gnx_age / gnx_move -> export_migrants + import -> export_halo +
 *   import_ghosts -> tile_pairs -> [all-gather pair_info, all-reduce bins 1] ->
 *   tile_offspring -> [requests -> serve_gametes -> put_gametes] ->
 *   tile_finish_births -> [all-reduce bins 0] -> tile_die.                      */
typedef struct {
  float x, y;
  int32_t age, sex;
  int64_t id;
  float fit;
  int32_t nbr_mask;   /* halo: bit (dy+1)*3+(dx+1) = neighbour tile that needs it */
} gnx_ind_rec;

int gnx_tile_set(gnx_state* h, int32_t R, int32_t C, int32_t r, int32_t c);
int gnx_tile_export_migrants(gnx_state* h, int64_t* n_out);
/* halo = whole hash cells: every individual whose cell lies within 2 cells of a
 * neighbour tile's cell range (complete candidate lists for the neighbour's own
 * focal individuals and for the ghosts they can choose)                        */
int gnx_tile_export_halo(gnx_state* h, int64_t* n_out);
int gnx_tile_get_staged(gnx_state* h, gnx_ind_rec* rec, float* z /*[n][n_traits]*/,
                        uint64_t* geno /*[n][2][W64]*/);
int gnx_tile_import(gnx_state* h, int64_t n, const gnx_ind_rec* rec, const float* z,
                    const uint64_t* geno);
int gnx_tile_import_ghosts(gnx_state* h, int64_t n, const gnx_ind_rec* rec);
int gnx_tile_pairs(gnx_state* h, int32_t burn, int64_t* n_pairs, int64_t* n_births);
int gnx_tile_pair_info(gnx_state* h, int64_t* focal_ids /*[P] order keys (cell << 40 | id), ascending*/,
                       int32_t* n_births /*[P]*/);
int gnx_density_bin_count(gnx_state* h);
/* which: 0 = individuals, 1 = pair midpoints; int32 [bin_count]             */
int gnx_get_bins(gnx_state* h, int32_t which, int32_t* out);
int gnx_set_bins(gnx_state* h, int32_t which, const int32_t* in);
int gnx_tile_offspring(gnx_state* h, int32_t burn, int64_t id_base,
                       const int64_t* pair_goff /*[P]*/, int64_t* n_requests);
int gnx_tile_get_requests(gnx_state* h, int64_t* parent_id, int32_t* child_k, int32_t* key,
                          uint8_t* start, float* px, float* py);
int gnx_tile_serve_gametes(gnx_state* h, int64_t n, const int64_t* parent_ids,
                           const int32_t* keys, const uint8_t* starts,
                           uint64_t* out /*[n][W64]*/);
int gnx_tile_put_gametes(gnx_state* h, int64_t n, const int32_t* child_k,
                         const uint64_t* data /*[n][W64]*/);
int gnx_tile_finish_births(gnx_state* h, int32_t burn);
int gnx_tile_die(gnx_state* h, int32_t burn, int32_t with_selection, int32_t have_pairs);
int gnx_set_max_id(gnx_state* h, int64_t max_id);

/* Device-resident transport: RCCL moves GPU memory, so under the "nccl" backend
 * the payloads never visit the host.  The *_dev entry points group the staged
 * selection by destination rank ON the device, return per-rank counts
 * (counts[R*C], the only thing that crosses PCIe) and DEVICE addresses that the
 * host layer wraps in tensors for isend/irecv; imports read the buffers RCCL
 * filled.  Returned addresses belong to the handle and stay valid until the
 * next tile call; every call returns with the handle's stream idle.          */
typedef struct {
  int64_t parent_id;
  int32_t key;      /* recombination path */
  int32_t start;    /* starting homologue, 0 / 1 */
} gnx_gamete_req;

int gnx_tile_export_migrants_dev(gnx_state* h, int64_t* counts /*[R*C]*/);
int gnx_tile_export_halo_dev(gnx_state* h, int64_t* counts /*[R*C]*/);
/* grouped selection: rec gnx_ind_rec[n], z float[n][n_traits], geno u64[n][2][W64] */
int gnx_tile_staged_ptrs(gnx_state* h, void** rec, void** z, void** geno);
int gnx_tile_import_dev(gnx_state* h, int64_t n, const void* rec, const void* z,
                        const void* geno);
int gnx_tile_import_ghosts_dev(gnx_state* h, int64_t n, const void* rec);
/* order keys (cell << 40 | focal id) int64[P] ascending; n_births int32[P] or NULL    */
int gnx_tile_pair_ptrs(gnx_state* h, int64_t* n_pairs, void** focal_ids, void** n_births);
/* the same without waiting for the handle's stream (tile2)                             */
int gnx_tile_pair_ptrs_nosync(gnx_state* h, int64_t* n_pairs, void** focal_ids, void** n_births);
int gnx_tile_offspring_dev(gnx_state* h, int32_t burn, int64_t id_base,
                           const void* pair_goff_dev /*int64[P]*/, int64_t* n_requests);
/* this step's gamete requests grouped by owner rank: gnx_gamete_req[n_requests] */
int gnx_tile_group_requests(gnx_state* h, int64_t* counts /*[R*C]*/, void** req_dev);
int gnx_tile_serve_gametes_dev(gnx_state* h, int64_t n, const void* req_dev,
                               void** out_dev /*u64[n][W64]*/);
/* answers in the grouped request order                                         */
int gnx_tile_put_gametes_dev(gnx_state* h, int64_t n, const void* data_dev);
/* int32 [2][bin_count]: individuals, pair midpoints (all-reduce in place)     */
int gnx_tile_bins_ptr(gnx_state* h, void** bins, int64_t* n_total);

/* ---- the device-driven tile protocol ("tile2"): the host waits for the device three
 * times per step - for the routing counts, for the pair / request counts and for the
 * survivor count - and every payload stays in device memory.  Entry points marked (no
 * wait) only enqueue work on the handle's stream (gnx_stream_ptr: the host layer orders
 * its collectives behind it with stream dependencies, never with a host synchronisation);
 * buffers the caller hands in must stay alive until gnx_tile2_die returns.
 *   gnx_tile2_move_route -> [counts all-gather, ONE batch of sends: migrants + ghosts] ->
 *   gnx_tile2_import -> gnx_tile2_pairs -> [counts all-gather, pair keys all-gather] ->
 *   gnx_tile2_offspring -> [requests] -> gnx_tile2_serve -> [gametes] -> gnx_tile2_put ->
 *   gnx_tile2_finish_births -> [ONE all-reduce: both bin fields + counters] -> gnx_tile2_die */
typedef struct {
  int64_t parent_id;
  int32_t key;      /* recombination path */
  int32_t start;    /* starting homologue, 0 / 1 */
  float px, py;     /* the parent's position (its hash cell locates it on the owning tile) */
} gnx_gamete_req2;
/* the HIP stream the handle enqueues on (torch.cuda.ExternalStream)                       */
int gnx_stream_ptr(gnx_state* h, void** stream);
/* age (+ movement), then the route of every individual: one that left the tile goes to
 * the tile that owns its new position, with its genome, and every individual in a hash
 * cell within two cells of a neighbour tile goes there as a ghost - emigrants included
 * (relative to their NEW tile; this tile can be among the receivers), so the halo does not
 * wait for the migrants' arrival.  Grouped by destination rank on the device.
 * counts[2 * R*C] = migrants per rank, ghosts per rank.  Emigrants stay in their slots
 * until the cell sort of gnx_tile2_pairs moves them behind the population.  One wait.     */
int gnx_tile2_move_route(gnx_state* h, int32_t move, int64_t* counts);
/* ... in two halves, for a caller that exchanges the counts itself (gnx_tile_step: the wait for
 * this tile's counts is the count exchange's): _begin enqueues age + movement + the counting
 * pass, *counts_dev = int32[2 * R*C] in device memory (no wait); _finish takes this tile's counts
 * from the host and enqueues the pass that fills the staging buffers (no wait).              */
int gnx_tile2_route_begin(gnx_state* h, int32_t move, void** counts_dev);
int gnx_tile2_route_finish(gnx_state* h, const int64_t* counts);
/* gnx_tile2_pairs leaves its gamete-request counts in device memory (*counts_dev, int32[R*C])
 * instead of waiting for them (on != 0); the caller hands them back once they have reached the
 * host with its own exchange (gnx_tile2_set_requests) before gnx_tile2_offspring.             */
int gnx_tile2_requests_dev(gnx_state* h, int32_t on, void** counts_dev);
int gnx_tile2_set_requests(gnx_state* h, const int64_t* req);
/* gnx_tile_step on several tiles, a fixed number of births per pair: gnx_tile2_pairs does not
 * wait for the pair count either (mode 1: counts[0] = counts[1] = -1) - the pairs' density bins
 * and the request counts read it on the device - and _settle does the host's bookkeeping once
 * the count has arrived with the caller's own exchange (births: fixed lambda x P).            */
int gnx_tile2_pairs_mode(gnx_state* h, int32_t nowait);
int gnx_tile2_pairs_settle(gnx_state* h, int32_t burn, int64_t n_pairs, int64_t* births);
/* the offspring that took a remote gamete re-read their alleles at the selected loci and their
 * phenotype from their finished rows (gnx_tile2_finish_births does it unless this has)        */
int gnx_tile2_settle_births(gnx_state* h, int32_t burn);
/* Tile-major offspring ids (gnx_set_id_order 1) through a caller-driven protocol
 * (TiledStepper._step_v2): counts[64] = this tile's births per virtual tile (one wait; the
 * classification rode with gnx_tile2_pairs), then bases[64] = the virtual tiles' global base
 * offsets in births (the exclusive sums of every tile's counts) before gnx_tile2_offspring,
 * which is then handed no offsets.                                                           */
int gnx_tile2_vt_counts(gnx_state* h, int64_t* counts);
int gnx_tile2_vt_bases(gnx_state* h, const int64_t* bases);
int gnx_tile2_route_ptrs(gnx_state* h, void** mig_rec, void** mig_z, void** mig_geno,
                         void** ghost_rec);
/* arrivals (device buffers): migrants rec / z / geno [n_mig], ghosts rec [n_ghost]; (no wait).
 * Capacity: the emigrants of this step still hold their slots and genome rows when the
 * arrivals are appended (they leave with the cell sort of gnx_tile2_pairs), so cap_inds must
 * hold residents + emigrants + immigrants + ghosts (+ the step's births) and cap_rows
 * residents + emigrants + immigrants; "capacity exceeded importing ..." otherwise.          */
int gnx_tile2_import(gnx_state* h, int64_t n_mig, const void* rec, const void* z,
                     const void* geno, int64_t n_ghost, const void* ghost_rec);
/* cell sort (emigrants leave, their genome rows return to the free stack), mate search, pair
 * list, births.  counts[2 + R*C] = pairs, births, gamete requests per owning rank.  One
 * wait (two more with Poisson-distributed births).                                      */
int gnx_tile2_pairs(gnx_state* h, int32_t burn, int64_t* counts);
/* offspring; *req_dev = gnx_gamete_req2[] grouped by owning rank as counted by
 * gnx_tile2_pairs; (no wait)                                                             */
int gnx_tile2_offspring(gnx_state* h, int32_t burn, int64_t id_base,
                        const void* pair_goff_dev /*int64[P]*/, void** req_dev);
/* gametes for n requests of other tiles: *out_dev = u64[n][W64]; (no wait)                */
int gnx_tile2_serve(gnx_state* h, int64_t n, const void* req_dev, void** out_dev);
/* the answers to this tile's requests, in the grouped request order; (no wait)          */
int gnx_tile2_put(gnx_state* h, int64_t n, const void* data_dev);
/* phenotypes of the offspring, bins of the tile's own individuals; *reduce_dev = int32
 * [2 * bin_count + 4]: both bin fields, then this tile's N, births, deaths of the PREVIOUS
 * step and 0 - one all-reduce(sum) in place for all of it; (no wait)                     */
int gnx_tile2_finish_births(gnx_state* h, int32_t burn, void** reduce_dev, int64_t* n_words);
/* densities from the reduced bins, death probabilities, mortality; checks what the
 * (no wait) calls could not report (records off the landscape, unknown parents).  One wait.
 * totals[3] = the all-reduced N, births, deaths-of-the-previous-step words.              */
int gnx_tile2_die(gnx_state* h, int32_t burn, int32_t with_selection, int32_t have_pairs,
                  int64_t* totals);

/* ---- one tiled time step per call, the exchanges issued by the library (csrc/gnx_comm.hip).
 * The tile2 protocol above composed in C: grouped ncclSend / ncclRecv to the neighbour tiles
 * on the handle's own stream (RCCL over xGMI), two KB-sized ncclAllGather for the counts (the
 * second carries the 64 virtual-tile birth counts the offspring ids are numbered from and the
 * pair count), ONE ncclAllReduce for both density fields and the counters - no torch.distributed
 * call, no Python between the phases of a step.  The reference has no counterpart (one process;
 * sim/model.py:924-925 is a TODO).
 *   gnx_comm_unique_id: rank 0 makes the id (ncclGetUniqueId, 128 bytes) and hands it to the
 *     other ranks by whatever channel the launcher has (bench.py: torch.distributed broadcast);
 *   gnx_comm_init_rccl: every rank joins (ncclCommInitRank; world 1 is allowed);
 *   gnx_comm_init_single: one tile, no communicator at all;
 *   gnx_comm_local_*: the tiles are handles of ONE process driven by one host thread each and
 *     the exchanges are device-to-device copies behind a barrier of the threads - the tests of
 *     the one-GPU box, where RCCL refuses two ranks on one device; everything but the nccl*
 *     calls themselves is the same code;
 *   gnx_tile_step: one step.  out[3]: exact != 0 -> the global (N after the step, births,
 *     deaths), one more KB-sized collective; else what rode on the step's own all-reduce:
 *     (N at the START of the step, births, deaths of the PREVIOUS step).  Poisson births
 *     (ops/mating.py:120-126) are numbered like fixed ones: the virtual tiles' BIRTH counts travel.
 *   gnx_tile_step_begin / _end: the same step in two calls, split where the reference's
 *     _do_pop_dynamics has its host work on the newborns - mutation (ops/mutation.py:169-206) and
 *     pedigree rows (structs/species.py:692-736) sit between the births and the deaths: after
 *     _begin every offspring of the step has its record, its alleles at the selected loci and
 *     its phenotype (gnx_last_births, gnx_mutate work there); gnx_tile_step_births tells the first
 *     id and the number of the step's offspring over ALL tiles; _end takes the densities' one
 *     all-reduce, the death probabilities and the mortality.  gnx_tile_step = both.
 *   gnx_comm_probe: librccl can be loaded and has every entry point the transport uses - what
 *     the ranks tell each other (over the launcher's CPU group) BEFORE any of them enters
 *     gnx_comm_init_rccl, whose ncclCommInitRank nobody can be called back from; the init itself
 *     carries a deadline (GNX_COMM_INIT_TIMEOUT_S, default 180 s): past it the rank says why on
 *     stderr and ends the process with code 86.
 *   The global maximum id (gnx_set_max_id) must be the same on every rank before the first
 *   step; the library keeps it.                                                             */
/* 0 (default): offspring ids in the canonical (hash cell, focal id) order of the pairs over the
 * whole landscape; 1: virtual tile by virtual tile (a fixed 8 x 8 blocking of the landscape, every
 * tile grid that divides it is a union of), inside one in that canonical order - what
 * gnx_tile_step hands out, since the tiles then only have to tell each other 64 birth counts
 * instead of every pair's order key; a one-device run in this order reproduces a tiled run id by
 * id.  The reference's own order is that of a Python set (ops/mating.py:63): unspecified.      */
int gnx_set_id_order(gnx_state* h, int32_t mode);
int gnx_comm_unique_id(uint8_t* out128);
int gnx_comm_init_rccl(gnx_state* h, const uint8_t* id128, int32_t rank, int32_t world);
int gnx_comm_init_single(gnx_state* h);
int gnx_comm_local_create(int32_t world, void** group);
int gnx_comm_local_join(gnx_state* h, void* group, int32_t rank);
int gnx_comm_local_abort(void* group);
int gnx_comm_local_destroy(void* group);
int gnx_comm_free(gnx_state* h);
int64_t gnx_comm_bytes_sent(gnx_state* h);
/* Known words through every operation of the handle's transport (the gather of host and device
 * words, a ragged two-part exchange with every rank including itself, the in-place sum), checked
 * on the host; collective: every rank of the communicator calls it.  Nonzero and gnx_last_error
 * when anything arrives wrong.  geonomics_amd.parallel.TiledStepper runs it once after the ranks
 * have joined.  (GNX_COMM_FORCE_RCCL=1 in the environment of gnx_comm_init_rccl makes a ONE-rank
 * communicator use the RCCL calls themselves instead of the one-rank shortcuts.) */
int gnx_comm_selftest(gnx_state* h);
int gnx_tile_step(gnx_state* h, int32_t burn, int32_t with_selection, int32_t exact, int64_t* out);
int gnx_tile_step_begin(gnx_state* h, int32_t burn);
int gnx_tile_step_births(gnx_state* h, int64_t* first_id, int64_t* total);
int gnx_tile_step_end(gnx_state* h, int32_t burn, int32_t with_selection, int32_t exact,
                      int64_t* out);
int gnx_comm_probe(void);
/* gnx_tile_step_abort: leave the state between _begin and _end without the density all-reduce and
 * the mortality - what every rank calls when the host's work on the newborns (the reference's
 * mutation / pedigree hooks, ops/mutation.py:169-206, structs/species.py:692-736) failed on ANY
 * rank, so that all of them raise instead of the others waiting inside the all-reduce.
 * gnx_comm_info: int64 out[16] - [0] transport (0 one rank, 1 RCCL, 2 local), [1] rank, [2] world,
 * [3] ncclCommCount, [4] ncclCommUserRank, [5] ncclCommCuDevice (-1 without an RCCL communicator),
 * [6] HIP device ordinal, [7] tiled steps taken, [8..12] host wall time of the step's phases summed
 * over them in microseconds (routing + count exchange; migrant / ghost exchange + import; cell
 * sort + pairs + second count exchange; births + gamete service; all-reduce + deaths), [13] bytes
 * sent, [14] collections of the genome blocks so far (also without a communicator).  No reference
 * counterpart (sim/model.py:924-925 is a TODO): it lets bench.py certify the ranks it ran on.    */
int gnx_tile_step_abort(gnx_state* h);
/* gnx_tile_walk: T tiled steps in one call with nothing between them (the tiles' gnx_walk; reference
 * Model.walk -> _do_timestep T times, sim/model.py:966-1161, on every rank).  Between two of its
 * steps the dead stay in their slots - no compaction: the next step's movement and routing skip
 * them, the imports go behind, the cell sort removes them with the emigrants - the last step
 * compacts.  out[5]: the last step's triple as gnx_tile_step reports it (exact as there), then
 * the sums over the T steps of the global population at the start of the step and of the births. */
int gnx_tile_walk(gnx_state* h, int64_t T, int32_t burn, int32_t with_selection, int32_t exact,
                  int64_t* out /*[5]*/);
int gnx_comm_info(gnx_state* h, int64_t* out /*[16]*/);

/* ---- pedigree (reference structs/species.py:692-736: rows of the tskit tables) --
 * the offspring of the last gnx_pop_dynamics_mate, in birth order; call it before
 * gnx_pop_dynamics_die.  keys / starts: recombination path and start homologue of
 * the gamete from parent 0 and from parent 1 (NULL to skip; burn-in has none)   */
int gnx_last_births(gnx_state* h, int64_t* child_id /*[B]*/, int64_t* parent_id /*[B][2]*/,
                    int32_t* keys /*[B][2]*/, uint8_t* starts /*[B][2]*/, float* xy /*[B][2]*/);

/* ---- statistics (reference sim/stats.py:359-435) ---------- */
/* per-locus count of 1-alleles over the 2N chromosomes and of heterozygous
 * individuals: het = cnt_het / N (_calc_het), f1 = cnt1 / 2N (_calc_maf)     */
int gnx_stats_locus_counts(gnx_state* h, int32_t* cnt1 /*[L]*/, int32_t* cnt_het /*[L]*/);
/* r^2 between the listed loci (_calc_ld); double [n][n], NaN on the diagonal */
int gnx_stats_ld(gnx_state* h, int32_t n_loci, const int32_t* loci, double* r2);
/* the counts behind r^2, which add over tiles: c[i] 1-alleles at locus i, cc[i][j]
 * chromosomes carrying 1 at both i and j                                        */
int gnx_stats_ld_counts(gnx_state* h, int32_t n_loci, const int32_t* loci, int64_t* c /*[n]*/,
                        int64_t* cc /*[n][n]*/);

/* ---- measurement ------------------------------------------------------------ */
int gnx_profiling(gnx_state* h, int32_t on);
/* accumulated HIP-event time (ms) and launch count of one kernel family,
 * measured on the handle's stream; resets the accumulator                   */
int gnx_kernel_time(gnx_state* h, int32_t kernel, double* ms, int64_t* launches,
                    double* algorithmic_bytes);
/* The box's own streaming rate, for the roofline line ("a measured
 * device-to-device copy"): a hand-written copy kernel, 16 bytes per lane and access,
 * grid-stride, over two buffers of `bytes` bytes each, `reps` launches timed with HIP
 * events on the current device; *gbps = (read + written bytes) / mean launch time.      */
int gnx_measure_copy(int64_t bytes, int32_t reps, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* GNX_HIP_H */
