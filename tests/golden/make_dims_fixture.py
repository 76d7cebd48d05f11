"""Builds tests/golden/env_dims.json: (env_name, robots, controller) -> (obs_dim, action_dim) for every run the
reference ships, read off the tensor SHAPES inside its params.pkl snapshots.

Nothing is unpickled or executed: the pickle stream (legacy torch format: the stream behind the magic / protocol /
sys-info pickles; zip format: <root>/data.pkl) is walked with pickletools.genops and only the size tuples that
follow a storage reference are collected.  policy.fc0 is (256, O), qf.fc0 is (256, O + A), policy.last_fc is (A, 256).

Run in the build container (needs /root/reference):  python tests/golden/make_dims_fixture.py
"""
import glob
import io
import json
import os
import pickletools
import zipfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _pickle_streams(path):
    if zipfile.is_zipfile(path):
        z = zipfile.ZipFile(path)
        name = next(n for n in z.namelist() if n.endswith("/data.pkl") or n == "data.pkl")
        yield z.read(name)
        return
    data = open(path, "rb").read()
    pos = 0
    for _ in range(4):                      # magic number, protocol version, sys info, the object
        f = io.BytesIO(data[pos:])
        for op, arg, off in pickletools.genops(f):
            if op.name == "STOP":
                end = off + 1
                break
        yield data[pos:pos + end]
        pos += end


def tensor_shapes(path):
    """2-D tensor sizes in stream order (ints pushed between a BINPERSID and the first TUPLE after it)."""
    shapes = []
    for stream in _pickle_streams(path):
        ints, armed = [], False
        for op, arg, _ in pickletools.genops(stream):
            if op.name == "BINPERSID":
                armed, ints = True, []
            elif armed and op.name in ("BININT", "BININT1", "BININT2", "LONG1"):
                ints.append(int(arg))
            elif armed and op.name in ("TUPLE1", "TUPLE2", "TUPLE3"):
                n = {"TUPLE1": 1, "TUPLE2": 2, "TUPLE3": 3}[op.name]
                if len(ints) >= n + 1:      # storage offset, then the size entries
                    shapes.append(tuple(ints[-n:]))
                armed = False
            elif armed and op.name not in ("BINGET", "LONG_BINGET", "BINPUT", "LONG_BINPUT", "MARK"):
                if op.name not in ("EMPTY_TUPLE",):
                    armed = armed       # other opcodes between the storage and its size do not occur
    return shapes


def dims_of(path):
    sh = [s for s in tensor_shapes(path) if len(s) == 2]
    ks = sorted({k for n, k in sh if n == 256 and k != 256})
    assert len(ks) == 2, (path, ks)
    O, A = ks[0], ks[1] - ks[0]
    assert (A, 256) in sh and (1, 256) in sh, (path, O, A)
    return O, A


def main():
    table = {}
    for vf in sorted(glob.glob(f"{REF}/runs/*/*/variant.json") + glob.glob(f"{REF}/log/runs/*/*/variant.json")):
        pk = os.path.join(os.path.dirname(vf), "params.pkl")
        if not os.path.exists(pk):
            continue
        e = json.load(open(vf))["expl_environment_kwargs"]
        robots = e["robots"] if isinstance(e["robots"], list) else [e["robots"]]
        key = "|".join([e["env_name"], "+".join(robots), e["controller"]])
        try:
            d = list(dims_of(pk))
        except Exception as ex:              # a snapshot without readable shapes: skip, others of the key cover it
            print("skip", pk, type(ex).__name__, ex)
            continue
        assert table.setdefault(key, d) == d, (key, table[key], d)
    with open(os.path.join(HERE, "env_dims.json"), "w") as f:
        json.dump(dict(source="tensor shapes inside /root/reference/{runs,log/runs}/*/*/params.pkl (pickletools walk, "
                              "nothing unpickled)", dims=table), f, indent=1, sort_keys=True)
    print(len(table), "keys")
    for k, v in sorted(table.items()):
        print(k, v)


if __name__ == "__main__":
    main()
