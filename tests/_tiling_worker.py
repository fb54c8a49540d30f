"""Worker for the tiling rehearsals: one process = one rank = one tile.

    python tests/_tiling_worker.py <backend> <shard> <world> <rank> <port> <steps> <out.npz>

shard = 'oracle' (numpy, CPU) or 'device' (libgnxhip.so; every rank on cuda:0).
Every rank builds the same global initial population, keeps its tile, steps, and
rank 0 writes the gathered final population."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)


def config():
    W, H, L = 64, 32, 192
    rasts = np.stack([np.ones((H, W)), np.tile(np.linspace(0, 1, W), (H, 1))]).astype(np.float32)
    rng = np.random.RandomState(3)
    import gnx_oracle as O
    paths = O.pack_bits(O.recomb_paths((rng.rand(24, L) < 0.02).astype(np.uint8)
                                       * (np.arange(L) > 0)))
    loci = np.array([5, 50, 100, 150])
    alpha = np.array([0.1, -0.1, 0.1, -0.1])
    return dict(W=W, H=H, L=L, rasts=rasts, paths=paths, loci=loci, alpha=alpha, N0=1500,
                seed=17, radius=3.0, K_factor=0.8, n_burn=3)


def make_oracle_shard(cfg, n_births_fixed=True):
    import gnx_step as S
    import gnx_shard as SH
    st = S.State(cfg['rasts'], S.Params(mating_radius=cfg['radius'], K_factor=cfg['K_factor'],
                                        n_births_fixed=n_births_fixed, n_births_lambda=1),
                 cfg['seed'], L=cfg['L'],
                 traits=[dict(loci=cfg['loci'], alpha=cfg['alpha'], layer=1, phi=0.05, gamma=1.0,
                              univ_adv=False)], paths_packed=cfg['paths'])
    st.init_population(cfg['N0'])
    return SH.OracleShard(st), st


def make_device_shard(cfg, n_births_fixed=True):
    from geonomics_amd import _native as nat
    from geonomics_amd.parallel import DeviceShard
    dev = nat.Device(cfg['W'], cfg['H'], 2, L=cfg['L'], n_traits=1, cap_inds=16384,
                     cap_rows=16384, seed=cfg['seed'], device=0)
    dev.upload_rasters(cfg['rasts'])
    dev.set_species_params(nat.default_species_params(
        mating_radius=-1.0 if cfg['radius'] is None else cfg['radius'], K_factor=cfg['K_factor'],
        n_births_fixed=int(n_births_fixed), n_births_lambda=1))
    dev.set_trait(0, cfg['loci'], cfg['alpha'], 1, 0.05, 1.0, False)
    dev.set_recomb_paths(cfg['paths'])
    dev.init_population(cfg['N0'])
    return DeviceShard(dev), dev


def run(backend, shard_kind, world, rank, port, steps, out, fixed=True, panmixia=False):
    from geonomics_amd.parallel import Comm, TiledStepper
    import gnx_oracle as O
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group(backend, rank=rank, world_size=world)
    comm = Comm(dist)
    cfg = config()
    if panmixia:
        # mating_radius None (reference structs/species.py:2178-2194): pairs drawn from everybody
        cfg['radius'] = None
    if shard_kind == 'oracle':
        shard, st = make_oracle_shard(cfg, fixed)
    else:
        shard, dev = make_device_shard(cfg, fixed)
    stepper = TiledStepper(shard, comm, cfg['W'], cfg['H'], cfg['radius'], move=True,
                           max_id=cfg['N0'] - 1, fixed_births=1 if fixed else 0,
                           use_library=False)     # (the gloo rehearsals: TiledStepper._step_v2)
    # keep only this tile's part of the common initial population
    stepper._migrate_initial = True
    rec, z, geno = shard.export_migrants()          # everybody outside my tile leaves
    hist = []
    for t in range(steps):
        burn = t < cfg['n_burn']
        if t == cfg['n_burn']:
            # genomes: the same global assignment on every rank, then keep own rows
            n_tot = comm.allreduce_sum(np.array([shard.counts()[0]], np.int64))[0]
            if shard_kind == 'oracle':
                ids = st.id.copy()
            else:
                from geonomics_amd import _native as nat
                ids = dev.download(nat.F_ID)
            all_ids = np.sort(np.concatenate(comm.allgather_i64(ids)))
            n_site = O.starting_mutation_counts(int(n_tot), np.full(cfg['L'], 0.5))
            G = O.starting_genomes(int(n_tot), cfg['L'], n_site, cfg['seed'])
            mine = G[np.searchsorted(all_ids, ids)]
            if shard_kind == 'oracle':
                st.set_genomes(mine)
            else:
                dev.upload_genomes(mine)
                shard.has_genomes = True
        hist.append(stepper.step(burn, not burn))
    # gather the final population on rank 0
    if shard_kind == 'oracle':
        ids, x, y, geno, zz = st.id, st.x, st.y, st.geno, st.z[:, 0]
        age = st.age
    else:
        from geonomics_amd import _native as nat
        ids, x, y = dev.download(nat.F_ID), dev.download(nat.F_X), dev.download(nat.F_Y)
        geno, zz, age = dev.download(nat.F_GENO), dev.download(nat.F_Z)[0], dev.download(nat.F_AGE)
    blob = np.concatenate([ids.astype(np.int64).view(np.uint8), x.view(np.uint8), y.view(np.uint8),
                           age.astype(np.int32).view(np.uint8),
                           zz.astype(np.float32).view(np.uint8),
                           np.ascontiguousarray(geno).view(np.uint8).ravel()])
    send = [blob if p == 0 else np.zeros(0, np.uint8) for p in range(world)]
    got = comm.alltoallv(send)
    if rank == 0:
        W64 = geno.shape[2]
        per = 8 + 4 + 4 + 4 + 4 + 16 * W64
        I, X, Y, A, Z, Gs = [], [], [], [], [], []
        for b in got:
            n = b.size // per
            o = 0
            I.append(b[o:o + 8 * n].view(np.int64)); o += 8 * n
            X.append(b[o:o + 4 * n].view(np.float32)); o += 4 * n
            Y.append(b[o:o + 4 * n].view(np.float32)); o += 4 * n
            A.append(b[o:o + 4 * n].view(np.int32)); o += 4 * n
            Z.append(b[o:o + 4 * n].view(np.float32)); o += 4 * n
            Gs.append(b[o:].view(np.uint64).reshape(n, 2, W64))
        I = np.concatenate(I)
        order = np.argsort(I)
        np.savez(out, ids=I[order], x=np.concatenate(X)[order], y=np.concatenate(Y)[order],
                 age=np.concatenate(A)[order], z=np.concatenate(Z)[order],
                 geno=np.concatenate(Gs)[order], hist=np.array(hist),
                 bytes_sent=stepper.bytes_sent, dev_transport=int(stepper.dev_transport))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    a = sys.argv
    run(a[1], a[2], int(a[3]), int(a[4]), int(a[5]), int(a[6]), a[7],
        fixed=(len(a) < 9 or a[8] in ('fixed', 'panmixia')),
        panmixia=(len(a) >= 9 and a[8] == 'panmixia'))
