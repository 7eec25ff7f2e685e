"""Pure-torch probe of hipGraph stream capture: a side stream that is forked from the capturing stream TWICE (event record on main + wait on side,
work, later a second record / wait, more work, then one join).  Dumps the captured DAG (hipGraphDebugDotPrint) so that the dependencies of the second
batch of side-stream nodes can be read off: does the second wait MERGE with the side stream's own last node or REPLACE it (leaving the first batch
unjoined, i.e. free to overlap whatever follows the graph launch)?"""
import os, sys, torch

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/capture_fork_twice.dot'
dev = torch.device('cuda:0')
a = torch.randn(512, 512, device=dev)
side = torch.cuda.Stream()
s = torch.cuda.Stream()
bufs = [torch.zeros(512, 512, device=dev) for _ in range(6)]


def issue():
    main = torch.cuda.current_stream()
    torch.mul(a, 2.0, out=bufs[0])                    # main node M0
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.add(bufs[0], 1.0, out=bufs[1])          # side node S0 (first fork)
    torch.mul(bufs[0], 3.0, out=bufs[2])              # main node M1
    ev2 = torch.cuda.Event(); ev2.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev2)
        torch.add(bufs[2], 1.0, out=bufs[3])          # side node S1 (second fork): must depend on S0 AND M1
    torch.mul(bufs[2], 5.0, out=bufs[4])              # main node M2
    ev3 = torch.cuda.Event(); ev3.record(side)
    main.wait_event(ev3)                              # join
    torch.add(bufs[4], bufs[3], out=bufs[5])          # main node M3: depends on M2 and S1


s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    issue()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    issue()
os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
g.debug_dump(out)
print('dumped', out)
print('dot exists:', os.path.exists(out), os.path.getsize(out) if os.path.exists(out) else 0)

# ---- functional check: a SLOW first side batch, a second side batch that consumes its result --------------------------------------------------------
big = torch.randn(4096, 4096, device=dev) * 0.01
x = torch.zeros(4096, 4096, device=dev); y = torch.zeros(4096, 4096, device=dev); z = torch.zeros(4096, 4096, device=dev)
m1 = torch.zeros(4096, 4096, device=dev); m2 = torch.zeros(4096, 4096, device=dev)
tmp = [torch.zeros(4096, 4096, device=dev) for _ in range(2)]


def issue2():
    main = torch.cuda.current_stream()
    torch.mul(big, 2.0, out=m1)
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.mm(m1, big, out=tmp[0])
        for i in range(30):                            # slow chain
            torch.mm(tmp[i % 2], big, out=tmp[(i + 1) % 2])
        torch.add(tmp[0], 1.0, out=x)
    torch.mul(m1, 3.0, out=m2)                          # quick main work
    ev2 = torch.cuda.Event(); ev2.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev2)
        torch.add(x, m2, out=y)                         # needs the slow chain (side order) AND m2 (second wait)
    ev3 = torch.cuda.Event(); ev3.record(side)
    main.wait_event(ev3)
    torch.mul(y, 1.0, out=z)


with torch.cuda.stream(s):
    issue2()
torch.cuda.synchronize()
ref = z.clone()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    issue2()
for rep in range(3):
    x.zero_(); y.zero_(); z.zero_(); m1.zero_(); m2.zero_(); tmp[0].zero_(); tmp[1].zero_()       # stale values of the previous replay must not hide a missing dependency
    torch.cuda.synchronize()
    g2.replay()
    x.fill_(float('nan'))                               # right after the launch on the same stream: any node that is not joined into the graph's end reads / writes late
    torch.cuda.synchronize()
    print('replay', rep, 'z == eager:', bool(torch.equal(z, ref)), 'max diff', float((z - ref).abs().max()))
