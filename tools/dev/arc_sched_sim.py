"""DEV: scheduling study of the Lee-Richards arc pass on the CPU.  Input: the dump of the emulation built with
-DLR2_EMU_DUMP (every tile's queue: item, list length, mask words).  Counts, per tile, the wave trips of the arc loop,
the merge steps and the refills under the shipped policy and under candidates.
   g++ ... -DSASA_EMU -DLR2_EMU_DUMP -include cstdio -include cstdlib -shared -o /tmp/libsasa_emu_dump.so tests/emu/emu.cpp
   python tools/dev/arc_sched_sim.py /tmp/dump.txt [cover_build=1]"""
import sys, math

C_ARC, C_MERGE, C_REFILL, C_ITER = 100, 75, 110, 10  # wave instructions: per arc trip, per merge step, per refill, per look at the waiting lanes

def tiles(path):
    cur = None
    for line in open(path):
        f = line.split()
        if f[0] == "T":
            if cur: yield cur
            cur = dict(nq=int(f[1]), na=int(f[2]), mwt=int(f[3]), cover=int(f[4]), items=[])
        else:
            nn = int(f[1]); mask = 0
            for k, w in enumerate(f[2:]): mask |= int(w, 16) << (32 * k)
            cur["items"].append((nn, mask))
    if cur: yield cur

def popc(m): return bin(m).count("1")
def rng(mask, lo, hi): return popc((mask >> lo) & ((1 << (hi - lo)) - 1)) if hi > lo else 0

def shared_fixed(t):
    nq = t["nq"]
    shb = 2 if nq * 4 <= 64 else 1
    share = 1 << shb
    trips = 0
    for nn, mask in t["items"]:
        for j in range(share):
            trips = max(trips, rng(mask, (nn * j) >> shb, (nn * (j + 1)) >> shb))
    return trips * C_ARC + shb * C_MERGE, trips

def queue(t, refill=24, steps=2):
    items = [popc(m) for _, m in t["items"]]
    nq = len(items)
    lanes = [items[i] if i < nq else 0 for i in range(64)]
    nxt = 64; cost = 0; its = 0; refills = 0
    while True:
        due = refill if nxt < nq else 64
        while True:
            waiting = sum(1 for w in lanes if w == 0)
            cost += C_ITER
            if waiting >= due: break
            its += 1
            for s in range(steps):
                if any(w > 0 for w in lanes): cost += C_ARC
                lanes = [w - 1 if w > 0 else 0 for w in lanes]
        refills += 1; cost += C_REFILL
        for l in range(64):
            if lanes[l] == 0 and nxt < nq: lanes[l] = items[nxt]; nxt += 1
        if nxt >= nq and all(w == 0 for w in lanes): break
    return cost, its * steps

def balanced(t, by_count=False, gmax=99):
    cnts = [popc(m) for _, m in t["items"]]
    if not cnts: return 0, 0, 0
    g = max(1, math.ceil(sum(cnts) / 64))
    while sum(math.ceil(c / g) for c in cnts) > 64: g += 1
    if g > gmax: return None
    trips = 0; nmax = 1
    for (nn, mask), c in zip(t["items"], cnts):
        n = math.ceil(c / g); nmax = max(nmax, n)
        if by_count: trips = max(trips, math.ceil(c / n))
        else:
            for k in range(n): trips = max(trips, rng(mask, nn * k // n, nn * (k + 1) // n))
    msteps = math.ceil(math.log2(nmax)) if nmax > 1 else 0
    return trips * C_ARC + msteps * C_MERGE + 40, trips, msteps  # (+40: choosing g, the lane table)

def main():
    path = sys.argv[1]; cover_build = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = 0; cur = 0; curtrips = 0; arcs = 0; nqs = 0
    bal = {False: 0, True: 0}; baltr = {False: 0, True: 0}; q_only = 0; nshared = 0; hyb = 0
    for t in tiles(path):
        n += 1; nqs += t["nq"]; arcs += sum(popc(m) for _, m in t["items"])
        if t["nq"] == 0: continue
        if cover_build and t["nq"] * 2 <= 64: c, tr = shared_fixed(t); nshared += 1
        else: c, tr = queue(t)
        cq, _ = queue(t)
        cur += c; curtrips += tr; q_only += cq
        for bc in (False, True):
            b = balanced(t, bc)
            bal[bc] += b[0]; baltr[bc] += b[1]
        hyb += min(balanced(t, False)[0], cq)
    print(f"tiles {n}  items with arcs per tile {nqs / n:.1f}  arcs per tile {arcs / n:.1f}  ideal trips {arcs / n / 64:.2f}  shared-path tiles {nshared}")
    print(f"shipped policy : {cur / n:7.0f} instr/tile, arc trips {curtrips / n:.2f}")
    print(f"queue only     : {q_only / n:7.0f}")
    print(f"balanced (pos) : {bal[False] / n:7.0f}, trips {baltr[False] / n:.2f}")
    print(f"balanced (cnt) : {bal[True] / n:7.0f}, trips {baltr[True] / n:.2f}")
    print(f"min(bal pos, queue) per tile: {hyb / n:7.0f}")

main()


def balanced_cap(t, ncap, c_arc=60, c_merge=75):
    """by arc count, at most ncap lanes per item"""
    cnts = [popc(m) for _, m in t["items"]]
    if not cnts: return 0
    g = max(1, math.ceil(sum(cnts) / 64))
    while sum(min(ncap, math.ceil(c / g)) for c in cnts) > 64: g += 1
    trips = 0; nmax = 1
    for c in cnts:
        n = min(ncap, math.ceil(c / g)); nmax = max(nmax, n)
        trips = max(trips, math.ceil(c / n))
    return trips * c_arc + (math.ceil(math.log2(nmax)) if nmax > 1 else 0) * c_merge

if len(sys.argv) > 3:
    ts = [t for t in tiles(sys.argv[1]) if t["nq"] > 0]
    for cap in (1, 2, 3, 4, 6, 8, 16, 64):
        print(f"lanes per item <= {cap:2d}: {sum(balanced_cap(t, cap) for t in ts) / len(ts):7.0f} instr/tile (arc trips x 60 + merge steps x 75)")
