"""A model check of the peer-store exchange's ordering argument: windows are single-buffered and exchange
numbers only grow — is a row ever overwritten before its reader is done, or read before it is complete?

The model has exactly the ordering the product has and nothing more:
  * every rank runs ONE stream: pack(e) -> stage1(e) -> stage2(e) -> combine(e) -> pack(e+1) -> ... (kernels of a stream do not overlap);
  * pack(e) of rank r writes the recv segment r of EVERY rank's window, then publishes recv_flag[r] = e there;
  * stage1(e) of rank r may start only when every recv_flag of r's window has reached e; it READS all recv segments of r's window;
  * stage2(e) of rank r writes the ret segment r of every rank's window, then publishes ret_flag[r] = e there;
  * combine(e) of rank r may start only when every ret_flag of its window has reached e; it READS all ret segments of its window.
A kernel is modelled as begin / end events with its writes and reads spread in between in random order, and a random scheduler
interleaves the ranks arbitrarily (any rank whose next event is enabled may move) — far more hostile than real timing.
Checked at every read: the segment holds the rows of THIS exchange (not an older one: read too early; not a newer one: overwritten
too early).  The same skeleton covers the routed and the broadcast form (who writes which segment when is identical)."""
import random


def _run(world, exchanges, seed):
    rng = random.Random(seed)
    recv_ver = [[0] * world for _ in range(world)]   # recv_ver[r][p]: exchange whose rows from p sit in r's window (0: none)
    ret_ver = [[0] * world for _ in range(world)]
    recv_flag = [[0] * world for _ in range(world)]
    ret_flag = [[0] * world for _ in range(world)]
    # per-rank program counter over micro-events of its stream
    progs = []
    for r in range(world):
        ev = []
        for e in range(1, exchanges + 1):
            w = [("w_recv", p, e) for p in range(world)]
            rng.shuffle(w)
            ev += w + [("pub_recv", p, e) for p in range(world)]              # pack: all stores drained, THEN the flags
            ev += [("wait_recv", e)]
            rd = [("r_recv", p, e) for p in range(world)]
            rng.shuffle(rd)
            ev += rd                                                            # stage 1 reads the rows
            w = [("w_ret", p, e) for p in range(world)] + [("r_recv", p, e) for p in range(world)]  # (the generic path's push kernel
            rng.shuffle(w)                                                      #  still reads the row tails while it stores outputs)
            ev += w + [("pub_ret", p, e) for p in range(world)]               # stage 2: stores drained, then the flags
            ev += [("wait_ret", e)]
            rd = [("r_ret", p, e) for p in range(world)]
            rng.shuffle(rd)
            ev += rd                                                            # combine reads the outputs
        progs.append(ev)
    pc = [0] * world
    steps = 0
    while any(pc[r] < len(progs[r]) for r in range(world)):
        movable = []
        for r in range(world):
            if pc[r] >= len(progs[r]):
                continue
            op = progs[r][pc[r]]
            if op[0] == "wait_recv" and min(recv_flag[r]) < op[1]:
                continue
            if op[0] == "wait_ret" and min(ret_flag[r]) < op[1]:
                continue
            movable.append(r)
        assert movable, f"deadlock at {pc} (world {world})"
        r = rng.choice(movable)
        op = progs[r][pc[r]]
        pc[r] += 1
        steps += 1
        kind = op[0]
        if kind == "w_recv":
            recv_ver[op[1]][r] = op[2]
        elif kind == "pub_recv":
            recv_flag[op[1]][r] = op[2]
        elif kind == "r_recv":
            assert recv_ver[r][op[1]] == op[2], f"rank {r} read rows of exchange {recv_ver[r][op[1]]} from rank {op[1]} while in exchange {op[2]}"
        elif kind == "w_ret":
            ret_ver[op[1]][r] = op[2]
        elif kind == "pub_ret":
            ret_flag[op[1]][r] = op[2]
        elif kind == "r_ret":
            assert ret_ver[r][op[1]] == op[2], f"rank {r} read outputs of exchange {ret_ver[r][op[1]]} from owner {op[1]} while in exchange {op[2]}"
    return steps


def test_single_buffered_windows_are_never_overwritten_early_nor_read_early():
    total = 0
    for world in (1, 2, 3, 4, 8):
        for seed in range(60 if world <= 4 else 10):
            total += _run(world, exchanges=6, seed=1000 * world + seed)
    assert total > 0


def test_the_model_does_catch_a_broken_protocol():
    """the same scheduler with ONE ordering rule removed — the flag published before the stores are drained — must fail: the check
    is not vacuous"""
    rng = random.Random(7)
    world = 2
    caught = 0
    for trial in range(200):
        recv_ver = [[0] * world for _ in range(world)]
        recv_flag = [[0] * world for _ in range(world)]
        # rank 0: publishes first, writes later (broken); rank 1: waits for the flag, then reads
        prog0, prog1 = [("pub", 1), ("w", 1)], [("wait", 1), ("r", 1)]
        pcs, ok = [0, 0], True
        while pcs[0] < 2 or pcs[1] < 2:
            mov = []
            if pcs[0] < 2:
                mov.append(0)
            if pcs[1] < 2 and not (prog1[pcs[1]][0] == "wait" and recv_flag[1][0] < 1):
                mov.append(1)
            r = rng.choice(mov)
            op = (prog0 if r == 0 else prog1)[pcs[r]]
            pcs[r] += 1
            if op[0] == "pub":
                recv_flag[1][0] = 1
            elif op[0] == "w":
                recv_ver[1][0] = 1
            elif op[0] == "r" and recv_ver[1][0] != 1:
                ok = False
        caught += not ok
    assert caught > 0
