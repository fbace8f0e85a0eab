set -e
cd $GRAFT_REPO_ROOT
rm -rf /tmp/u2a /tmp/u2b && mkdir -p /tmp/u2a /tmp/u2b
(cd /tmp/u2a && python $GRAFT_REPO_ROOT/importance_generation.py --net u2netp --dataset DUTS --synthetic --input_size 288 --batch_size 3 --limit 1 --single_sweep --deferred > log.txt 2>&1; tail -1 log.txt)
(cd /tmp/u2b && python $GRAFT_REPO_ROOT/importance_generation.py --net u2netp --dataset DUTS --synthetic --input_size 288 --batch_size 3 --limit 1 --single_sweep --device_accumulate > log.txt 2>&1; tail -1 log.txt)
python - <<'PY'
import numpy as np, glob, os
a=sorted(glob.glob('/tmp/u2a/importance_score/*/*.npy')); b=sorted(glob.glob('/tmp/u2b/importance_score/*/*.npy'))
print(len(a), len(b))
worst=0
for fa,fb in zip(a,b):
    assert os.path.basename(fa)==os.path.basename(fb)
    x,y=np.load(fa),np.load(fb)
    assert x.shape==y.shape
    d=np.abs(x-y)/np.maximum(np.abs(y),1e-30)
    worst=max(worst,float(d.max()))
print('max rel diff deferred(batched multi) vs per-hook:', worst)
PY
