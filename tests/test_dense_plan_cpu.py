"""Static schedule of the one-launch direct solver (mov-slam_amd/csrc/dense_plan.{h,cpp}; the kernel is dense_persist.hip):
replayed on the CPU against its own flags.  The solver stands in for the reference's LinearSolverCSparse factorisation
(/root/reference/src/Optimizer.cc:535); on the device every workgroup runs its task list in order and waits for flags other
workgroups set, so the list must (1) cover the blocked Cholesky exactly, (2) never wait for a flag nobody sets, (3) always
leave some workgroup able to run — whatever the speeds of the workgroups."""
import numpy as np
import pytest

ASM, UPD, DIAG, OFF, RHS, BSX, BSC, EPI, RUP, UPD2, COL = range(11)


def waits_and_sets(op, I, K, k, p0, p1, nt):
    """(flags a task waits for, flag it sets)"""
    if op == ASM: return [], []
    if op == UPD:       # operands the workgroup owns itself (p0 / p1 = their LDS slots) are not waited for: program order;
        # nor are those the update before left in the scratch tiles (-2)
        return ([("F", I, k)] if p0 == -1 else []) + ([("F", K, k)] if I != K and p1 == -1 else []), []
    if op == UPD2: return ([("F", K, k)] if p0 < 0 else []) + ([("F", K - 1, k)] if p1 < 0 else []), []
    if op == COL: return [("PD", K - 1)], [("F", K, K - 1), ("PD", K)]
    if op == DIAG: return [], [("PD", K)]
    if op == OFF: return [("PD", K)], [("F", I, K)]
    if op == RHS: return ([("FY", K - 1)] if K >= 1 else []), [("FY", K)]
    if op == RUP: return [("FY", k)], []
    if op == BSX: return [("FY", K)] + [("FC", i, K) for i in range(K + 1, nt)], [("FX", K)] + ([("FC", K, K - 1)] if K >= 1 else [])
    if op == BSC: return ([] if p0 == 1 else [("FX", I)]), [("FC", I, K)]
    if op == EPI: return [("FX", j) for j in range(nt)], []
    raise AssertionError(op)


def replay(plan, nt, order_rng):
    """run the workgroups' lists in a random interleaving; returns the number of tasks executed"""
    tp, tk = plan["task_ptr"], plan["tasks"]
    pc = tp[:-1].copy()
    flags, done = set(), 0
    upd_seen = {}                         # tile -> columns applied so far (must be 0, 1, 2, ... in order)
    state = {}                            # tile -> "asm" / "final"
    slots = {}                            # (workgroup, slot) -> tile
    rup_seen = {}                         # block row -> columns applied to its right-hand side row by RUP tasks
    scratch = {}                          # workgroup -> [tile in scratch A, tile in scratch B] as the device leaves them
    while True:
        runnable = [g for g in range(plan["G"]) if pc[g] < tp[g + 1] and all(f in flags for f in waits_and_sets(*tk[pc[g], [0, 2, 3, 4, 5, 6]], nt)[0])]
        if not runnable:
            break
        g = int(order_rng.choice(runnable))
        op, slot, I, K, k, p0, p1 = (int(v) for v in tk[pc[g]])
        p2 = int(plan["tasks8"][pc[g], 7])
        tile = (I, K)
        if op == ASM:
            assert tile not in state and slots.setdefault((g, slot), tile) == tile
            state[tile] = "asm"; upd_seen[tile] = 0
        elif op == UPD:
            assert state[tile] == "asm" and upd_seen[tile] == k and slots[(g, slot)] == tile
            upd_seen[tile] += 1
            sc = scratch.setdefault(g, [None, None])
            # an operand marked "still in scratch" must be what the device has there; both operands must be final tiles
            if p0 == -1: sc[0] = (I, k)
            elif p0 == -2: assert sc[0] == (I, k)
            else: assert slots[(g, p0)] == (I, k)
            if I != K:
                if p1 == -1: sc[1] = (K, k)
                elif p1 == -2: assert sc[1] == (K, k)
                else: assert slots[(g, p1)] == (K, k)
            assert state[(I, k)] == "final" and state[(K, k)] == "final"
        elif op == UPD2:
            # the diagonal tile and the tile to its left take block column k together (k <= K - 2)
            sub = (K, K - 1)
            assert I == K == g and k <= K - 2 and slots[(g, slot)] == tile and slots[(g, p2)] == sub
            assert state[tile] == "asm" and state[sub] == "asm" and upd_seen[tile] == k and upd_seen[sub] == k
            upd_seen[tile] += 1; upd_seen[sub] += 1
            assert state[(K, k)] == "final" and state[(K - 1, k)] == "final"
            sc = scratch.setdefault(g, [None, None])
            if p0 < 0: sc[0] = (K, k)
            else: assert slots[(g, p0)] == (K, k)
            if p1 < 0: sc[1] = (K - 1, k)
            else: assert slots[(g, p1)] == (K - 1, k)
        elif op == COL:
            # the diagonal owner's chain of block column K - 1: its sub-diagonal tile finished, the diagonal tile's last update, W_K
            sub = (K, K - 1)
            assert I == K == g and K >= 1 and slots[(g, slot)] == tile and slots[(g, p2)] == sub
            assert state[sub] == "asm" and upd_seen[sub] == K - 1 and state[tile] == "asm" and upd_seen[tile] == K - 1
            upd_seen[tile] += 1
            state[sub] = "final"; state[tile] = "final"
            scratch.setdefault(g, [None, None])[1] = None
        elif op in (DIAG, OFF):
            assert state[tile] == "asm" and upd_seen[tile] == K and slots[(g, slot)] == tile    # every column to the left applied
            state[tile] = "final"
            if op == OFF: scratch.setdefault(g, [None, None])[1] = None                         # (D_K is fetched into scratch B)
        elif op == RUP:
            assert g == K and rup_seen.get(K, 0) == k and k <= K - 2 and state[(K, k)] == "final" and upd_seen[(K, K)] > k
            rup_seen[K] = k + 1
            scratch.setdefault(g, [None, None])[0] = None
        elif op in (RHS, BSX):
            assert state[(K, K)] == "final" and slots[(g, slot)] == (K, K)
            if op == BSX:                            # ... and the contribution of the owner's own sub-diagonal tile
                assert K == 0 or (slots[(g, p0)] == (K, K - 1) and state[(K, K - 1)] == "final")
            if op == RHS:
                assert rup_seen.get(K, 0) == max(K - 1, 0)       # RHS itself applies block column K - 1, from the workgroup's own tile
                assert K == 0 or (slots[(g, p0)] == (K, K - 1) and state[(K, K - 1)] == "final")
                scratch.setdefault(g, [None, None])[0] = None
        elif op == BSC:
            assert state[tile] == "final" and slots[(g, slot)] == tile
        for s in waits_and_sets(op, I, K, k, p0, p1, nt)[1]:
            assert s not in flags
            flags.add(s)
        pc[g] += 1; done += 1
    assert (pc == tp[1:]).all(), "schedule stalls: some workgroup waits for a flag that is never set"
    return done, state


@pytest.mark.parametrize("nt", [1, 2, 3, 7, 10, 19, 24, 50])
def test_schedule_covers_the_factorisation_and_never_stalls(built_lib, nt):
    plan = built_lib.dense_plan(nt)
    assert plan["ok"] and plan["G"] <= 248 and plan["slots"] <= 6
    for seed in range(3 if nt <= 24 else 1):
        done, state = replay(plan, nt, np.random.default_rng(seed))
        assert done == len(plan["tasks"])
        assert set(state) == {(I, K) for K in range(nt) for I in range(K, nt)} and set(state.values()) == {"final"}
    # the chain D_(K-1) -> L(K, K-1) -> D_K stays inside one workgroup: the diagonal tile's owner owns the tile to its left
    tk, tp = plan["tasks"], plan["task_ptr"]
    owner = {}
    for g in range(plan["G"]):
        for op, slot, I, K, k, p0, p1 in tk[tp[g]:tp[g + 1]]:
            if op == ASM:
                owner[(int(I), int(K))] = g
    for K in range(1, nt):
        assert owner[(K, K)] == owner[(K, K - 1)] == K
    assert int((tk[:, 0] == EPI).sum()) == 1 and owner[(0, 0)] == 0


def test_systems_beyond_the_slots_of_one_launch_are_refused(built_lib):
    assert not built_lib.dense_plan(60)["ok"]                   # 1 830 tiles > 248 workgroups x 6 slots
    assert not built_lib.dense_plan(8, max_groups=4)["ok"]      # more block columns than workgroups
    small = built_lib.dense_plan(8, max_groups=12, max_slots=4)
    assert small["ok"] and small["G"] == 12 and small["slots"] <= 4
    replay(small, 8, np.random.default_rng(1))
