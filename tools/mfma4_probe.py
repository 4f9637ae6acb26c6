"""Reads the dump of tools/mfma4_probe (lines "A la B lb D l": output lane l is non-zero when A is 1 at lane la and B
is 1 at lane lb) and prints which lane bits carry (block, i, k) of A, (block, k, j) of B and (block, i, j) of D for
v_mfma_f64_4x4x4_4b_f64: D_blk[i][j] += sum_k A_blk[i][k] B_blk[k][j]."""
import sys
from collections import defaultdict

pairs = defaultdict(list)
for ln in open(sys.argv[1]):
    w = ln.split()
    if len(w) == 6 and w[0] == "A":
        pairs[(int(w[1]), int(w[3]))].append(int(w[5]))
print("non-zero (la, lb) pairs:", len(pairs), " outputs per pair:", sorted({len(v) for v in pairs.values()}))
# for A lane la, the set of B lanes that combine with it share (block, k); the output lane is determined by (block, i, j)
partners = defaultdict(set)
for (la, lb) in pairs:
    partners[la].add(lb)


def bits(x):
    return "".join(str((x >> k) & 1) for k in range(5, -1, -1))


print("A lane -> B partner lanes (same block and k):")
for la in (0, 1, 2, 3, 4, 8, 12, 16, 32):
    print(f"  A {la:2d} ({bits(la)}): B {sorted(partners[la])}")
print("output lane for (A lane, B lane):")
for la, lb in ((0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (4, 4), (4, 5), (5, 4), (16, 16), (17, 18), (32, 32), (48, 48)):
    print(f"  A {la:2d} B {lb:2d} -> D {pairs.get((la, lb))}")
