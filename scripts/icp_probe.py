"""GPU probe: per-iteration phase profile of the ICP kernel on the bench workload."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np
from kiss_icp_amd import _cabi
from kiss_icp_amd.config import load_config
from kiss_icp_amd.datasets import generate_scans, kitti_like, kitti_like_vegetated, livox_like
from kiss_icp_amd.kiss_icp import KissICP
opts = dict(a.split('=') for a in sys.argv[1:])
street = opts.pop('street', '0') == '1'  # street=1: round 1's bare street scene
livox = opts.pop('livox', '0') == '1'    # livox=1: the 1M-point / 0.1 m configuration
frames = int(opts.pop('frames', '0'))    # frames=N: drive N frames first (default 30 / 20 / 14)
csv = opts.pop('csv', '')                 # csv=PATH: one row per workgroup of the last launch (for fitting the run weights)
opts.setdefault('icp_profile', '1')
for k, v in opts.items():
    _cabi.set_option(k, int(v))
nf = frames or (14 if street else (20 if livox else 30))
scans = generate_scans(livox_like if livox else (kitti_like if street else kitti_like_vegetated), dict(seed=2 if livox else 0, n_frames=nf), range(nf))
k = KissICP(load_config(deskew=False, voxel_size=0.1) if livox else load_config(deskew=False))
for i in range(nf):
    k.register_frame_async(scans[i][0])
k.sync()
prof = k.icp_iteration_profile()
print(opts, k.icp_profile(), 'n_src', k.last_stats()['n_source'])
cyc, tk = k.icp_clock()
print('last launch: %d shader cycles in %.2f us -> %.0f MHz' % (cyc, tk / 100, cyc / max(tk, 1) * 100))
print(' it  assoc publish gather solve | max_assoc passes   (us)')
for i, r in enumerate(prof):
    print('%3d %6.2f %6.2f %6.2f %6.2f | %6.2f %4d' % (i, r[0] / 100, r[1] / 100, r[2] / 100, r[3] / 100, r[4] / 100, r[5]))

gp = k.icp_group_profile()
if gp.size and not livox:
    # group form: which groups searched at all (path 5 = none: every point of the group's workgroup kept its neighbour, or the
    # other groups took the few that did not; its "examined" field then carries the workgroup's number of searches)
    print(' it  groups searching  workgroups with a search  most searches in one workgroup')
    for it in range(gp.shape[0]):
        g = gp[it].reshape(-1, 16, gp.shape[2])
        idle = g[:, :, 6] == 5
        per_wg = np.where(idle.any(axis=1), np.where(idle, g[:, :, 5], 0).max(axis=1), 16)  # (16: every group searched: at least that many)
        print('%3d %8d %18d %22d' % (it, int((~idle).sum()), int(((~idle).any(axis=1)).sum()), int(per_wg.max())))
if gp.size:
    names = ('wait-in', 'xform', 'fill', 'scan')
    for it in (0, 1, min(6, gp.shape[0] - 1), gp.shape[0] - 1):
        g = gp[it]
        tot = g[:, 0:4].sum(axis=1)
        print('iteration %d: groups %d  paths %s' % (it, g.shape[0], np.bincount(g[:, 6], minlength=4).tolist()))
        for j, nm in enumerate(names):
            print('   %-8s mean %6.2f  p50 %6.2f  p99 %6.2f  max %6.2f us' % (
                nm, g[:, j].mean() / 100, np.percentile(g[:, j], 50) / 100, np.percentile(g[:, j], 99) / 100, g[:, j].max() / 100))
        if it > 0 and (g[:, 6] == 1).any():
            # group form, later iterations: the 'xform' field of a searching group carries its scan-list BUILD (0: the list stood)
            sg = g[g[:, 6] == 1]
            b = sg[:, 1] > 0
            print('   searches by scan list: %d, of them with a list build %d (build mean %.2f max %.2f us; search incl. build mean %.2f max %.2f); without: search mean %.2f max %.2f us' % (
                len(sg), int(b.sum()), sg[b, 1].mean() / 100 if b.any() else 0, sg[b, 1].max() / 100 if b.any() else 0,
                sg[b, 3].mean() / 100 if b.any() else 0, sg[b, 3].max() / 100 if b.any() else 0,
                sg[~b, 3].mean() / 100 if (~b).any() else 0, sg[~b, 3].max() / 100 if (~b).any() else 0))
        gc = g.reshape(-1, 16, g.shape[1])[:, [0, 10], :].reshape(-1, g.shape[1]) if (g[:, 6] == 4).all() else g  # (thread-per-query form: the other groups' records carry counters)
        print('   staged   mean %6.1f  max %d   examined mean %6.1f max %d' % (gc[:, 4].mean(), gc[:, 4].max(), gc[:, 5].mean(), gc[:, 5].max()))
        if (g[:, 6] == 4).all():
            # thread-per-query form: the 'xform' field of a record carries the voxels its thread visited {LDS | map << 8}
            # (the 'wait-in' field is phase A); in iteration 0 the records of groups 0..4 carry the window phase's parts instead
            gw = g.reshape(-1, 16, g.shape[1])
            ev = gw[:, (6 if it == 0 else 0)::2, :].reshape(-1, g.shape[1])  # even groups: visits
            od = gw[:, (5 if it == 0 else 1)::2, :].reshape(-1, g.shape[1])  # odd groups: walk | lookup, chains
            vis = ev[:, 1]
            print('   voxels per query: walked by the thread mean %.2f max %d, filed as items mean %.2f max %d' % (
                (vis & 255).mean(), (vis & 255).max(), (vis >> 8).mean(), (vis >> 8).max()))
            for nm, col in (('lookups', od[:, 4]), ('chains', od[:, 5]), ('walk', od[:, 1])):
                print('   wave %-8s mean %6.2f  p50 %6.2f  p99 %6.2f  max %6.2f us' % (nm, col.mean() / 100, np.percentile(col, 50) / 100, np.percentile(col, 99) / 100, col.max() / 100))
            wgd, wgr, wgi, wgm, wgt = gw[:, 2, 4], gw[:, 2, 5], gw[:, 4, 4], gw[:, 4, 5], gw[:, :, 8].max(axis=1) / 100.0
            print('   per workgroup: map-direct queries mean %.1f max %d; items served mean %.1f max %d (in the map mean %.1f max %d) in rounds mean %.2f max %d' % (
                wgd.mean(), wgd.max(), wgi.mean(), wgi.max(), wgm.mean(), wgm.max(), wgr.mean(), wgr.max()))
            tf, ts, tm, tc = gw[:, 6, 4] / 100.0, gw[:, 6, 5] / 100.0, gw[:, 8, 4] / 100.0, gw[:, 8, 5] / 100.0
            print('   per workgroup (us): filing + wait mean %.2f max %.2f; serving mean %.2f max %.2f; merging mean %.2f max %.2f; phase C mean %.2f max %.2f' % (
                tf.mean(), tf.max(), ts.mean(), ts.max(), tm.mean(), tm.max(), tc.mean(), tc.max()))
            nfull, nlook, tlook, comp = gw[:, 12, 4], gw[:, 12, 5], gw[:, 14, 4] / 100.0, gw[:, 14, 5] & 1
            print('   per workgroup: full searches mean %.1f max %d (of %d queries at most); of them with table lookups mean %.1f max %d; slowest lookups + chains mean %.2f max %.2f us; compacted in %d of %d workgroups' % (
                nfull.mean(), nfull.max(), gw[:, 0, 7].max(), nlook.mean(), nlook.max(), tlook.mean(), tlook.max(), comp.sum(), len(comp)))
            print('   correlation of the workgroup time with: full searches %.2f  searches with lookups %.2f  slowest lookups %.2f  items %.2f' % (
                np.corrcoef(wgt, nfull)[0, 1], np.corrcoef(wgt, nlook)[0, 1], np.corrcoef(wgt, tlook)[0, 1], np.corrcoef(wgt, wgi)[0, 1]))
            order = np.argsort(-wgt)
            for w in list(order[:8]) + list(order[len(order) // 2: len(order) // 2 + 4]):
                print('      wg %3d: %6.2f us  run %3d  direct %3d  items %4d (map %4d)  rounds %d   file %5.2f serve %5.2f merge %5.2f C %5.2f   full %3d lookups %3d slowest %5.2f%s' % (
                    w, wgt[w], gw[w, 0, 7], wgd[w], wgi[w], wgm[w], wgr[w], tf[w], ts[w], tm[w], tc[w], nfull[w], nlook[w], tlook[w], ' compact' if comp[w] else ''))
            slow = np.argsort(-od[:, 3])[:6]
            for w in slow:
                print('   slow wave: scan %6.2f = lookups %6.2f + chains %6.2f + walk %6.2f us (+ queueing)' % (od[w, 3] / 100, od[w, 4] / 100, od[w, 5] / 100, od[w, 1] / 100))
        if it == 0 and int(opts.get('icp_bulk_fill', 1)):
            ph = g.reshape(-1, 16, g.shape[1])[:, 0:5, 1] / 100.0  # groups 0..4 carry tile_fill_bulk's phases in the xform field (4: the lookups alone)
            for j, nm in enumerate(('windows+dedup', 'enter (+lookups)', 'fetch', 'verdicts', 'lookups alone')):
                print('   bulk fill %-14s mean %6.2f  p50 %6.2f  max %6.2f us' % (nm, ph[:, j].mean(), np.percentile(ph[:, j], 50), ph[:, j].max()))
            gw = g.reshape(-1, 16, g.shape[1])
            tot_w = ph[:, 0:4].sum(axis=1) + ph[:, 4]
            for w in np.argsort(-tot_w)[:8]:
                print('      wg %3d: dedup %5.2f lookups %5.2f enter %5.2f fetch %5.2f verdicts %5.2f us   run %3d points, tile %4d points' % (
                    w, ph[w, 0], ph[w, 4], ph[w, 1], ph[w, 2], ph[w, 3], gw[w, 0, 7], gw[w, :, 4].max()))
            print('      correlation of the fill time with run points %.2f, with tile points %.2f' % (
                np.corrcoef(tot_w, gw[:, 0, 7])[0, 1], np.corrcoef(tot_w, gw[:, :, 4].max(axis=1))[0, 1]))
        worst = np.argsort(-tot)[:6]
        for w in worst:
            print('   slow group %4d (wg %3d): wait %5.2f xform %5.2f fill %5.2f scan %5.2f us staged %4d examined %4d path %d' % (
                w, w // 16, g[w, 0] / 100, g[w, 1] / 100, g[w, 2] / 100, g[w, 3] / 100, g[w, 4], g[w, 5], g[w, 6]))

    # per workgroup (iteration 6 or the last): points of its run, points in its tile, search time of its slowest group
    it = min(6, gp.shape[0] - 1)
    g = gp[it].reshape(-1, 16, 9)
    wg_t = g[:, :, 8].max(axis=1) / 100.0
    wg_n = g[:, 0, 7]
    clean = [0, 10] if (gp[it][:, 6] == 4).all() else list(range(16))
    wg_staged = g[:, clean, 4].max(axis=1)
    wg_ex = g[:, clean, 5].mean(axis=1)
    used = wg_n > 0
    print('per workgroup, iteration %d: %d workgroups with points; search time of the slowest group: mean %.2f  p50 %.2f  p90 %.2f  max %.2f us' % (
        it, used.sum(), wg_t[used].mean(), np.percentile(wg_t[used], 50), np.percentile(wg_t[used], 90), wg_t[used].max()))
    print('   correlation of that time with: run points %.2f   tile points %.2f   examined per first point %.2f' % (
        np.corrcoef(wg_t[used], wg_n[used])[0, 1], np.corrcoef(wg_t[used], wg_staged[used])[0, 1], np.corrcoef(wg_t[used], wg_ex[used])[0, 1]))
    order = np.argsort(-wg_t)
    for w in list(order[:12]) + list(order[used.sum() // 2: used.sum() // 2 + 4]) + list(order[used.sum() - 6: used.sum()]):
        print('   wg %3d: %6.2f us  run %4d points  tile %5d points  examined/first point %5.1f' % (w, wg_t[w], wg_n[w], wg_staged[w], wg_ex[w]))

    if csv:
        # per workgroup, a later iteration: what a weight rule can see (run points, tile points, examined points) against
        # what it is meant to equalise (search time of the slowest group, window phase of the first iteration)
        it = min(6, gp.shape[0] - 1)
        g6, g0 = gp[it].reshape(-1, 16, 9), gp[0].reshape(-1, 16, 9)
        with open(csv, 'w') as f:
            f.write('wg,run_points,tile_points,examined_mean,examined_max,search_us_slowest_group,search_us_mean_group,fill0_us\n')
            for w in range(g6.shape[0]):
                if g6[w, 0, 7] == 0:
                    continue
                f.write('%d,%d,%d,%.1f,%d,%.2f,%.2f,%.2f\n' % (w, g6[w, 0, 7], g6[w, :, 4].max(), g6[w, :, 5].mean(), g6[w, :, 5].max(),
                                                              g6[w, :, 8].max() / 100.0, g6[w, :, 8].mean() / 100.0, g0[w, :, 2].max() / 100.0))
        print('wrote', csv)
