"""Diagnostic: how many branches of a captured hipGraph actually run concurrently on this runtime?
k branches of m spin kernels each (torch.cuda._sleep: one thread, fixed cycles) forked from one root and
joined; replay time ~ m*t if the branches overlap, k*m*t if they are serialised.  Also a "diamond in a
branch" shape like the neck's (a branch that forks again after its first node).

    python tools/diag_graph_parallel.py
"""
import time

import torch


def timed(g, n=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def capture(build):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        build()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    return g


def main():
    cyc = 200000                               # ~0.1 ms per spin kernel
    x = torch.zeros(8, device='cuda')
    streams = [torch.cuda.Stream() for _ in range(6)]

    def fan(k, m):
        def build():
            cur = torch.cuda.current_stream()
            x.add_(1)
            for s in streams[:k - 1]:
                s.wait_stream(cur)
            for b in range(k):
                ctx = torch.cuda.stream(streams[b - 1]) if b else torch.cuda.stream(cur)
                with ctx:
                    for _ in range(m):
                        torch.cuda._sleep(cyc)
            for s in streams[:k - 1]:
                cur.wait_stream(s)
            x.add_(1)
        return build

    base = timed(capture(fan(1, 4)))
    print(f'1 branch x 4 spins: {base:.3f} ms')
    for k in (2, 3, 4, 6):
        t = timed(capture(fan(k, 4)))
        print(f'{k} branches x 4 spins: {t:.3f} ms  (x{t / base:.2f} of one branch)')

    # the neck's shape: root -> B (side 0) ; root -> P (cur) -> {C (side 1), A2 (cur, also waits for B's first node)}
    def neck_like():
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        sB.wait_stream(cur)
        sC.wait_stream(cur)
        with torch.cuda.stream(sB):
            torch.cuda._sleep(cyc)
            ev = torch.cuda.Event()
            ev.record(sB)
            for _ in range(3):
                torch.cuda._sleep(cyc)
        torch.cuda._sleep(cyc)                  # P
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            for _ in range(4):
                torch.cuda._sleep(cyc)
        cur.wait_event(ev)
        for _ in range(3):
            torch.cuda._sleep(cyc)              # A2
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        x.add_(1)
    t = timed(capture(neck_like))
    print(f'neck-like (B 4 | P 1 -> C 4 | A2 3): {t:.3f} ms  (x{t / base * 4:.2f} spins; ideal 5, serial 12)')

    def spins(n):
        for _ in range(n):
            torch.cuda._sleep(cyc)

    def v1():      # nested fork, no cross edge: root -> B4 ; root -> P1 -> {C4, A2 3}
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(4)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(4)
        spins(3)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        x.add_(1)

    def v3():      # cross edge only: root -> B(1, event, 3) ; root -> P1 -> wait(event) -> A2 3
        cur = torch.cuda.current_stream()
        sB = streams[0]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(1)
            ev = torch.cuda.Event()
            ev.record(sB)
            spins(3)
        spins(1)
        cur.wait_event(ev)
        spins(3)
        cur.wait_stream(sB)
        x.add_(1)

    def v4():      # nested fork alone: root -> P1 -> {C4, A2 3}
        cur = torch.cuda.current_stream()
        sC = streams[1]
        x.add_(1)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(4)
        spins(3)
        cur.wait_stream(sC)
        x.add_(1)

    def v5():      # neck-like, but A2 on its own stream forked after P (cur idles until the join)
        cur = torch.cuda.current_stream()
        sB, sC, sA = streams[0], streams[1], streams[2]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(1)
            ev = torch.cuda.Event()
            ev.record(sB)
            spins(3)
        spins(1)
        sC.wait_stream(cur)
        sA.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(4)
        with torch.cuda.stream(sA):
            sA.wait_event(ev)
            spins(3)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        cur.wait_stream(sA)
        x.add_(1)

    def v6():      # P duplicated at the head of two root-level branches is not an option; instead fork everything at
                   # the root and make C wait for P through an event
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        sB.wait_stream(cur)
        sC.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(4)
        spins(1)
        evp = torch.cuda.Event()
        evp.record(cur)
        with torch.cuda.stream(sC):
            sC.wait_event(evp)
            spins(4)
        spins(3)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        x.add_(1)

    def v7():      # the event wait moved in front of P; B joins A2 mid-way while C still runs
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(1)
            ev = torch.cuda.Event()
            ev.record(sB)
            spins(3)
        cur.wait_event(ev)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(6)
        spins(3)
        cur.wait_stream(sB)
        spins(2)
        cur.wait_stream(sC)
        x.add_(1)

    def v8():      # B's first node on cur before any fork; B joins A2 mid-way while C still runs
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        spins(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(3)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(6)
        spins(3)
        cur.wait_stream(sB)
        spins(2)
        cur.wait_stream(sC)
        x.add_(1)

    def v9():      # like v8 but B joins only at the very end (its consumers run after the last join)
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        spins(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(3)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(6)
        spins(3)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        spins(2)
        x.add_(1)

    def v10():     # long chains of short kernels (40 x ~20 us): is the cross-stream signal sent right after its node?
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        short = cyc // 5
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            torch.cuda._sleep(short)
            ev = torch.cuda.Event()
            ev.record(sB)
            for _ in range(40):
                torch.cuda._sleep(short)
        cur.wait_event(ev)
        torch.cuda._sleep(short)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            for _ in range(40):
                torch.cuda._sleep(short)
        for _ in range(40):
            torch.cuda._sleep(short)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        x.add_(1)
    t = timed(capture(v10))
    print(f'v10 three chains of 40 short kernels: {t:.3f} ms  (x{t / base * 20:.1f} short kernels; ideal 42, serial 122)')

    def v11():     # two chains that exchange one event each way in the middle (no fork node at all)
        cur = torch.cuda.current_stream()
        sB = streams[0]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(2)
            ev_b = torch.cuda.Event()
            ev_b.record(sB)
        spins(2)
        ev_h = torch.cuda.Event()
        ev_h.record(cur)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_h)
            spins(3)
        cur.wait_event(ev_b)
        spins(3)
        cur.wait_stream(sB)
        x.add_(1)

    def v12():     # v11 plus a third chain C forked after the first node of cur (the neck's strand C)
        cur = torch.cuda.current_stream()
        sB, sC = streams[0], streams[1]
        x.add_(1)
        sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            spins(2)
            ev_b = torch.cuda.Event()
            ev_b.record(sB)
        spins(1)
        sC.wait_stream(cur)
        with torch.cuda.stream(sC):
            spins(5)
        spins(1)
        ev_h = torch.cuda.Event()
        ev_h.record(cur)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_h)
            spins(3)
        cur.wait_event(ev_b)
        spins(3)
        cur.wait_stream(sB)
        cur.wait_stream(sC)
        x.add_(1)

    for name, fn, ideal, serial in (('v7 event wait before P, mid join', v7, 8, 16), ('v8 B1 before the fork, mid join', v8, 8, 16),
                                    ('v9 B1 before the fork, joins at the end', v9, 10, 16),('v1 nested fork + B, no cross edge', v1, 5, 12), ('v3 cross edge only', v3, 4, 8),
                                    ('v4 nested fork alone', v4, 5, 8), ('v5 A2 on its own stream', v5, 5, 12),
                                    ('v6 C forked at root, waits P by event', v6, 5, 12),
                                    ('v11 two chains crossing in the middle', v11, 5, 10),
                                    ('v12 = v11 + a third chain forked after the first node', v12, 6, 15)):
        t = timed(capture(fn))
        print(f'{name}: {t:.3f} ms  (x{t / base * 4:.2f} spins; ideal {ideal}, serial {serial})')


if __name__ == '__main__':
    main()
