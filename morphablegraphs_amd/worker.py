"""A worker rank of the sharded product seam ("rank 0 drives; other ranks are workers").

One process per GPU, started as a FRESH process (before anything has touched a GPU):

    RANK=r WORLD_SIZE=N LOCAL_RANK=r MG_RDV_BASE=$(mktemp -d)/rdv python -m morphablegraphs_amd.worker --zip graph.zip      (a fresh private directory per run)
    ... or --synthetic-graph 16 / --synthetic-walk for the synthetic models of bench.py and the tests

Every rank loads the same model, joins the communicator (the RCCL unique id travels through files under MG_RDV_BASE) and sits
in distributed.worker_loop: it executes the commands rank 0 broadcasts -- score this block of these candidates, draw and score
your rows of this draw, run your share of this planner step -- until rank 0 sends {"op": "stop"}.  Rank 0 is the process
that runs the reference's graph-walk control flow with a communicator in its algorithm configuration
(HipMotionPrimitiveGenerator / evaluate_samples_using_constraints / HipPrimitiveSet.evaluate_options_on_device).
The reference's own process model is one process per core, each with its own graph
(examples/mg_rest_interface_parallel.py:252-254); nothing in its graph-walk code is sharded.
"""
import argparse
import os
import sys


def load_nodes(args, ctx):
    """{key: node} as rank 0 holds them, plus "__primitive_set__" (planner steps) and "__skeleton__"."""
    from . import _capi, synthetic
    from .motion_state_graph import HipMotionStateGraph, HipMotionStateGraphNode, HipPrimitiveSet
    nodes = {}
    if args.zip:
        graph = HipMotionStateGraph(context=ctx)
        graph.load_from_zip(args.zip)
        nodes.update(graph.nodes)
    elif args.synthetic_graph:
        prims = synthetic.make_graph_primitives(args.synthetic_graph)
        pset = HipPrimitiveSet(prims, context=ctx)
        nodes.update(pset.nodes)
        nodes["__primitive_set__"] = pset
    else:
        node = HipMotionStateGraphNode(context=ctx)
        node.init_from_dict("walk", {"name": "leftStance", "mm": synthetic.make_walk_primitive(seed=0)})
        nodes[node.node_key] = node
    if args.synthetic_skeleton:
        joints, animated = synthetic.make_skeleton()
        nodes["__skeleton__"] = _capi.Skeleton(joints, animated)
    return nodes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--zip", default=None, help="motion-state-graph zip (the reference's model file)")
    ap.add_argument("--synthetic-graph", type=int, default=0, help="n synthetic primitives (bench.py's graph configuration)")
    ap.add_argument("--synthetic-walk", action="store_true", help="the synthetic 'walk' primitive as node ('walk', 'leftStance')")
    ap.add_argument("--synthetic-skeleton", action="store_true")
    ap.add_argument("--transport", choices=("rccl", "files"), default="rccl",
                    help="rccl: mg_dist_* over xGMI (one GPU per rank); files: everything through the rendezvous files (rehearsal: ranks may share a GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK)")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", str(rank)))
    from . import _capi, distributed
    rdv = distributed.FileRendezvous(rank, world, base=os.environ.get("MG_RDV_BASE"))
    ctx = _capi.Context(device)            # raises when there is no GPU: no CPU fallback
    comm = distributed.open_communicator(ctx, rank, world, rdv, transport=args.transport)   # (rank 0 opens its own the same way)
    nodes = load_nodes(args, ctx)
    if rank == 0:
        raise SystemExit("rank 0 is the driver (the process that runs the graph walk), not a worker")
    try:
        served = distributed.worker_loop(comm, nodes)     # a rank that waits longer than the rendezvous' timeout raises: the process exits non-zero
    finally:
        if hasattr(comm, "close"):
            comm.close()
        rdv.cleanup()
    print("worker rank %d served %d commands" % (rank, served), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
