"""Debug aid: run one parity case repeatedly on both schedules and show where the fused sweeps'
aggregated volume S differs from the per-direction schedule (non-deterministic differences =
a race or a missed hardware hazard)."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import parity_util as U
from stereo_reconstruction_cv_amd import synth
H,W,D,bs,minD,mode,seed = 20,1100,512,3,0,1,11
if len(sys.argv) > 1: H,W,D,bs,minD,mode,seed = map(int, sys.argv[1:8])
debug = int(sys.argv[8]) if len(sys.argv) > 8 else 0
l,r,_ = synth.make_pair(H,W,D,seed)
p = U.params(D, bs, minD, mode, speckleWindowSize=30, speckleRange=2)
h0 = U.run_hip_with_taps(l,r,p,schedule=0)
S0=h0['S']
import bruteforce_sgbm as B
q = B.normalise(**p)
C64 = h0['C'].astype(np.int64)
Ldirs = {d: B.aggregate_path(C64, d[0], d[1], q['P1'], q['P2']) for d in B.DIRS8}
S_pass1 = sum(Ldirs[d] for d in B.DIRS8[:4])   # -> , down-right, down, down-left
S_pass2 = sum(Ldirs[d] for d in B.DIRS8[4:])
assert np.array_equal(np.minimum(S_pass1 + S_pass2, 32767), S0.astype(np.int64)), 'bruteforce S != schedule-0 S'
print(S0.shape)
nbad=0
for trial in range(6):
    for sr in (0,1,3):
        h = U.run_hip_with_taps(l,r,p,schedule=1,sweep_rows=sr,debug=debug)
        S1=h['S']
        bad=np.argwhere(S1!=S0)
        if not len(bad): continue
        nbad+=1
        pix=sorted(set((int(a),int(b)) for a,b,_ in bad))
        print('trial',trial,'rows',sr,'bad pixels',pix[:10], 'disp equal', np.array_equal(h['disp'],h0['disp']))
        y,x=pix[0]
        ds=bad[(bad[:,0]==y)&(bad[:,1]==x)][:,2]
        print('  d',ds.tolist()[:6], 'n', len(ds))
        for name, part in (('pass1', S_pass1), ('pass2', S_pass2)):
            print('  hip ==', name, ':', bool(np.array_equal(S1[y,x,ds].astype(np.int64), part[y,x,ds])), end=';')
        for d in B.DIRS8:
            if np.array_equal(S0[y,x,ds].astype(np.int64) - S1[y,x,ds], Ldirs[d][y,x,ds]): print(' missing = L', d, end='')
        print()
        U_ = S1[y,x,ds].astype(np.int64) - S_pass2[y,x,ds]   # the "S of pass 1" that was effectively added
        for name, vol in (('pass1', S_pass1), ('final', S0.astype(np.int64)), ('hipfinal', S1.astype(np.int64))):
            hit = np.argwhere((vol[:, :, ds] == U_[None, None, :]).all(axis=2))
            if len(hit): print('  effective Sp ==', name, 'at', hit.tolist()[:4], end=';')
        V_ = S1[y,x,ds].astype(np.int64) - S_pass1[y,x,ds]   # or: the pass-2 sum that was effectively added
        hit = np.argwhere((S_pass2[:, :, ds] == V_[None, None, :]).all(axis=2))
        if len(hit): print('  effective pass-2 sum == pass2 at', hit.tolist()[:4], end=';')
        miss = S0[y,x,ds].astype(np.int64) - S1[y,x,ds]
        if nbad <= 3:
            import os
            os.makedirs('/root/repo/gpurun_out', exist_ok=True)
            np.savez('/root/repo/gpurun_out/dbg_fail%d.npz' % nbad, y=y, x=x, ds=ds, hip=S1[y,x], ref=S0[y,x], sr=sr,
                     L=np.stack([Ldirs[d][y,x] for d in B.DIRS8]), C=C64[y,x],
                     Lnb=np.stack([Ldirs[d][max(y-1,0):y+2, max(x-1,0):x+2] for d in B.DIRS8]))
        for dX in B.DIRS8[4:]:
            for dY in B.DIRS8[4:]:
                if np.array_equal(miss, Ldirs[dY][y,x,ds] + Ldirs[dX][y,x].min()): print('  missing = L', dY, '+ min L', dX, end=';')
        for d in B.DIRS8[4:]:
            for d2 in B.DIRS8[4:]:
                if d < d2 and np.array_equal(S0[y,x,ds].astype(np.int64) - S1[y,x,ds], Ldirs[d][y,x,ds] + Ldirs[d2][y,x,ds]): print(' missing = L', d, '+ L', d2, end='')
        print()
        print('  hip', S1[y,x,ds[:6]].tolist(), 'ref', S0[y,x,ds[:6]].tolist(), 'hex', [hex(int(v)&0xffff) for v in S1[y,x,ds[:6]]])
print('runs with mismatches:',nbad,'of 18')
