# randomized matrices through the ICholT dataflow kernel and the ILUT wave kernel, bit-exact against the C restatement
#   python profiles/tools/fuzz_kernels.py [nseeds]
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fuzz_util
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sys.exit(1 if fuzz_util.run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, first_seed=first, verbose=2 if len(sys.argv) > 3 else True) else 0)
