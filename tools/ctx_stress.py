import sys, time
sys.path.insert(0, '/root/repo')
from circuits_amd import lib
L = lib()
t = time.time()
for i in range(80):
    c = L.ctx("rollup-main", nTx=4, nLevels=16, maxL1Tx=2, maxFeeTx=2, flags=2)
    c.close()
    if i % 10 == 9: print(i + 1, "contexts created and destroyed, %.1f s" % (time.time() - t), flush=True)
