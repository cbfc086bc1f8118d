"""HIP-graph capture topology probe: which fork/join patterns between side streams survive
hipStreamEndCapture on this ROCm (A/D: nested fork -> crash; B/C: forks from the origin + one cross edge -> ok;
E/F: forked stream waits on another forked stream and is waited back -> crash).  python tools/capture_topology.py A"""
import sys, torch, faulthandler
faulthandler.enable()
case = sys.argv[1]
x = torch.ones(1 << 20, device="cuda")
s1, k1 = torch.cuda.Stream(), torch.cuda.Stream()
cap = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        main = torch.cuda.current_stream()
        y = x * 2
        if case == "A":
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                a = y + 1
                k1.wait_stream(s1)
                with torch.cuda.stream(k1):
                    b = a * 3
                c = a + 2
                s1.wait_stream(k1)
                d = b + c
            main.wait_stream(s1)
        elif case in ("B", "C"):
            s1.wait_stream(main); k1.wait_stream(main)
            with torch.cuda.stream(k1):
                b = y * 3
            with torch.cuda.stream(s1):
                c = y + 2
                s1.wait_stream(k1)
                d = b + c
            main.wait_stream(s1)
            if case == "C":
                main.wait_stream(k1)
        elif case == "D":   # A, but k1 also joined to origin
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                a = y + 1
                k1.wait_stream(s1)
                with torch.cuda.stream(k1):
                    b = a * 3
                c = a + 2
                s1.wait_stream(k1)
                d = b + c
            main.wait_stream(s1)
            main.wait_stream(k1)
        elif case == "E":   # both forked from origin, then s1 -> k1 -> s1 edges
            s1.wait_stream(main); k1.wait_stream(main)
            with torch.cuda.stream(s1):
                a = y + 1
                k1.wait_stream(s1)
                with torch.cuda.stream(k1):
                    b = a * 3
                c = a + 2
                s1.wait_stream(k1)
                d = b + c
            main.wait_stream(s1)
        elif case == "F":   # E with an op on k1 before the s1 -> k1 edge
            s1.wait_stream(main); k1.wait_stream(main)
            with torch.cuda.stream(k1):
                w = y * 5
            with torch.cuda.stream(s1):
                a = y + 1
                k1.wait_stream(s1)
                with torch.cuda.stream(k1):
                    b = a * 3 + w
                c = a + 2
                s1.wait_stream(k1)
                d = b + c
            main.wait_stream(s1)
        z = d + 1
g.replay(); torch.cuda.synchronize()
print(case, "ok", float(z[0]))
