"""Discrete check of the strip pipeline of mtp_amd/csrc/gemm_s8.hip: the counted waits (RAW) and the slot reuse (WAR) under the
one-barrier stagger of the two wave groups, for a stream of NT K-tiles.  Same method as the model that checked gemm_p8.h in round 2.

Model: group 0 runs the read part R_g of phase g in interval 2g and its MFMAs in 2g+1, group 1 one interval later; an interval ends at
a barrier.  A wave's loads complete in issue order; `s_waitcnt vmcnt(N)` in R_g guarantees every load except the N youngest.  Adversary:
a load is taken to land as LATE as the waits allow when it is read (RAW) and as EARLY as its issue when it overwrites (WAR).
usage: python tools/probes/s8_schedule_check.py [k_tiles_total]"""
import sys

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 40      # K-tiles of the flattened stream
VM_EVEN, VM_ODD, VM_PRO = 12, 11, 11
EXTRA = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # side loads per even phase (ahead of the DMA): they must only make waits stricter

def triple_first(t):   # pieces in issue order
    return [("B0", t, 0), ("B0", t, 1), ("A", t, 0)]
def triple_second(t):
    return [("A", t, 1), ("B1", t, 0), ("B1", t, 1)]

# per-wave program: list of (phase, [reads], [issues], vm) ; phase -1 = prologue
prog = []
pro = triple_first(0) + triple_second(0) + triple_first(1) + triple_second(1) + triple_first(2)
prog.append((-1, [], pro, VM_PRO))
for T in range(NT):
    prog.append((2 * T, [("A", T), ("B0", T)], [("side", T, i) for i in range(EXTRA)] + triple_second(T + 2), VM_EVEN))
    prog.append((2 * T + 1, [("B1", T)], triple_first(T + 3), VM_ODD))

def slot(kind, t):
    return (kind, t % 3)

errors = 0
for group in (0, 1):
    # interval of R part of phase g for this group
    def r_interval(g, grp):
        return -1 if g < 0 else 2 * g + grp
    # when is each piece guaranteed landed (interval index whose closing barrier follows the wait), per group; issue interval too
    for grp_issuer in (0, 1):
        queue = []          # issued loads in order
        guaranteed = {}     # piece -> interval of the wait that covers it
        issued_at = {}
        for (g, reads, issues, vm) in prog:
            ti = r_interval(g, grp_issuer)
            for pc in issues:
                queue.append(pc)
                issued_at[pc] = ti
            done = queue[: max(0, len(queue) - vm)]
            for pc in done:
                guaranteed.setdefault(pc, ti)
        # a side load issued at the start of even phase 2T must be covered by the wait of odd phase 2T + 3 (where its slice runs)
        if group == grp_issuer:
            for pc, ti in issued_at.items():
                if pc[0] == "side" and pc[1] + 2 < NT:
                    need = r_interval(2 * pc[1] + 3, grp_issuer)
                    if guaranteed.get(pc, 1 << 30) > need:
                        print("side-load violation:", pc, "guaranteed at", guaranteed.get(pc), "needed in", need)
                        errors += 1
        # RAW: a read by `group` in interval tr needs every piece of the element, issued by waves of grp_issuer, guaranteed in an interval < tr
        for (g, reads, issues, vm) in prog:
            tr = r_interval(g, group)
            for (kind, t) in reads:
                for i in (0, 1):
                    pc = (kind, t, i)
                    if pc not in guaranteed or guaranteed[pc] >= tr:
                        print("RAW violation: group %d reads %s in interval %d, group-%d pieces guaranteed at %s" % (group, pc, tr, grp_issuer, guaranteed.get(pc)))
                        errors += 1
        # WAR: a piece issued by grp_issuer in interval ti into slot s may land at once: every read of the slot's previous content (by `group`)
        # must have retired in an interval < ti
        last_read = {}
        for (g, reads, issues, vm) in prog:
            tr = r_interval(g, group)
            for (kind, t) in reads:
                last_read[(kind, t)] = tr
        for pc, ti in issued_at.items():
            kind, t, i = pc
            if kind == "side" or t < 3:
                continue
            prev = (kind, t - 3)
            if prev in last_read and last_read[prev] >= ti:
                print("WAR violation: group %d issues %s in interval %d, group %d reads %s in interval %d" % (grp_issuer, pc, ti, group, prev, last_read[prev]))
                errors += 1
            if prev not in last_read and t - 3 < NT:
                print("?? no read of", prev)
print("checked %d K-tiles, %d side loads per even phase: %s" % (NT, EXTRA, "OK" if not errors else "%d violations" % errors))
# tightness: the largest vm constants that still pass would be found by raising VM_*; report the flight (intervals) each element gets
sys.exit(1 if errors else 0)
