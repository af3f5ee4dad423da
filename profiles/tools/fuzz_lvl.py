# randomized matrices through the level-ordered ILU(0) kernel and sweeps, IChol0 and ILUC, bit-exact against the C restatement
#   python profiles/tools/fuzz_lvl.py [nseeds [first]]
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fuzz_lvl
sys.exit(1 if fuzz_lvl.run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, first_seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
