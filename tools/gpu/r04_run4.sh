# round 4: a few schedule knobs of the tiled path at 1000 x 1000
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export HIPMF_MID_FRONT=0
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'relerr %.1e' % d['relative_error'])"; }
run default
HIPMF_UPD_G4=256 run g4_256
HIPMF_UPD_G4=512 run g4_512
HIPMF_UPD_G4=1024 run g4_1024
HIPMF_UPD_G4=512 HIPMF_UPD_G8=1024 run g4_512_g8_1024
HIPMF_OVERLAP_SMALL=0 run no_overlap
HIPMF_ARENA_REUSE=0 run no_arena_reuse
