"""Diagnostic: what this ROCm's stream capture accepts (each case in a fresh process: python tools/graph_min_probe.py CASE)."""
import sys, torch
case = sys.argv[1]
a = torch.zeros(1 << 20, device="cuda"); b = torch.zeros(1 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if case == "fork_join":
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        s1.wait_stream(cap); s2.wait_stream(cap)
        with torch.cuda.stream(s1): a.add_(1)
        with torch.cuda.stream(s2): b.add_(2)
        cap.wait_stream(s1); cap.wait_stream(s2)
        a.add_(b)
elif case == "events_churn":      # many events created, recorded, waited once or never, and dropped while capturing
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        s1.wait_stream(cap)
        for i in range(50):
            with torch.cuda.stream(s1): a.add_(1)
            ev = s1.record_event()
            if i % 2: cap.wait_event(ev)
            b.add_(1)
            s1.wait_event(cap.record_event())
            del ev
        cap.wait_stream(s1)
elif case == "event_waited_twice":
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        s1.wait_stream(cap); s2.wait_stream(cap)
        with torch.cuda.stream(s1): a.add_(1)
        ev = s1.record_event()
        cap.wait_event(ev); s2.wait_event(ev)
        with torch.cuda.stream(s2): b.add_(1)
        cap.wait_stream(s1); cap.wait_stream(s2)
elif case == "memset":
    with torch.cuda.graph(g):
        a.zero_(); b[100:200].zero_(); a.add_(1)
elif case == "unjoined":
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        s1.wait_stream(cap)
        with torch.cuda.stream(s1): a.add_(1)
elif case.startswith("v"):
    c = torch.zeros(1 << 20, device="cuda")
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        s1.wait_stream(cap); s2.wait_stream(cap)
        if case == "v1":      # the side stream waits for both chains once
            with torch.cuda.stream(s1): a.add_(1)
            b.add_(1)
            s2.wait_event(s1.record_event()); s2.wait_event(cap.record_event())
            with torch.cuda.stream(s2): c.add_(1)
        elif case == "v2":    # s1 -> s2 -> cap
            with torch.cuda.stream(s1): a.add_(1)
            s2.wait_event(s1.record_event())
            with torch.cuda.stream(s2): c.add_(1)
            cap.wait_event(s2.record_event())
            b.add_(1)
        elif case == "v3":    # both chains feed the side stream alternately, no edges back
            for st in (s1, cap, s1, cap):
                with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
                s2.wait_event(st.record_event())
                with torch.cuda.stream(s2): c.add_(1)
        elif case == "v4":    # ... and each chain later waits for the side stream's work on its behalf
            evs = []
            for st in (s1, cap, s1, cap):
                with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
                s2.wait_event(st.record_event())
                with torch.cuda.stream(s2):
                    c.add_(1)
                    evs.append((st, s2.record_event()))
            for st, ev in evs:
                st.wait_event(ev)
                with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
        elif case == "v5":    # v4 with a single round
            evs = []
            for st in (s1, cap):
                with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
                s2.wait_event(st.record_event())
                with torch.cuda.stream(s2):
                    c.add_(1)
                    evs.append((st, s2.record_event()))
            for st, ev in evs:
                st.wait_event(ev)
                with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
        elif case == "v6":    # only the non-origin chain gets an edge back from the side stream
            with torch.cuda.stream(s1): a.add_(1)
            s2.wait_event(s1.record_event())
            with torch.cuda.stream(s2): c.add_(1)
            ev = s2.record_event()
            b.add_(1)
            s1.wait_event(ev)
            with torch.cuda.stream(s1): a.add_(1)
        elif case == "v10":     # the ORIGIN plays the weight-gradient stream; both chains are forked streams with edges to and from it only
            evs = []
            for rnd in range(3):
                for st in (s1, s2):
                    with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
                    cap.wait_event(st.record_event())
                c.add_(1)                                   # the joint product of both chains' operands
                ev = cap.record_event()
                for st in (s1, s2):
                    with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
                evs.append(ev)
                if rnd >= 1:                                # a block later each chain waits before it overwrites what the origin read
                    for st in (s1, s2):
                        st.wait_event(evs[rnd - 1])
                        with torch.cuda.stream(st): (a if st is s1 else b).add_(1)
        elif case in ("v7", "v8", "v9"):
            s3 = torch.cuda.Stream()
            with torch.cuda.stream(s1): a.add_(1)
            s2.wait_event(s1.record_event())
            with torch.cuda.stream(s2): c.add_(1)
            ev = s2.record_event()
            b.add_(1)
            if case == "v7":       # the edge back goes through the origin stream
                cap.wait_event(ev)
                s1.wait_event(cap.record_event())
            elif case == "v8":     # ... through a helper stream that runs nothing
                s3.wait_stream(cap)
                s3.wait_event(ev)
                s1.wait_event(s3.record_event())
            elif case == "v9":     # the side stream is forked off the chain it serves instead of the origin
                pass
            with torch.cuda.stream(s1): a.add_(1)
            if case == "v8":
                cap.wait_stream(s3)
        elif case in ("v6a", "v6b", "v6c", "v6d"):
            with torch.cuda.stream(s1): a.add_(1)
            s2.wait_event(s1.record_event())
            with torch.cuda.stream(s2): c.add_(1)
            ev = s2.record_event()
            if case != "v6d": b.add_(1)
            s1.wait_event(ev)
            with torch.cuda.stream(s1): a.add_(1)
            if case == "v6b":
                with torch.cuda.stream(s2): c.add_(1)
            if case == "v6c":        # the side stream joins through the chain it last fed, not directly
                pass
        if case == "v6a":
            cap.wait_stream(s2); cap.wait_stream(s1)
        elif case == "v6c":
            cap.wait_stream(s1)
        else:
            cap.wait_stream(s1); cap.wait_stream(s2)
g.replay(); torch.cuda.synchronize()
print("OK", case, float(a[0]), float(b[0]))
