"""Checkpoints of a training run: raw fp32 blobs (.npy, no pickling) + a JSON manifest.

The reference saves `params.pkl` = torch.save of whole rlkit modules every epoch
(/root/reference/util/rlkit_custom.py:68-82, consumed by rlkit_utils.py:173-174,241-250) and nothing a run
could resume from: no optimizer state, entropy coefficient, step counters, replay buffer or sampling
stream (SURVEY.md 8b "Snapshot contract").  save_checkpoint() keeps all of that, so load_checkpoint()
continues a run bit for bit (tests/test_gpu_checkpoint.py).

Interop with the reference's own files, both without rlkit and without unpickling anything:
  * read_rlkit_zip_params(path): the fp32 storages `*/data/N` of a zip-format params.pkl -> flat nn.Linear
    vectors of policy / qf1 / qf2 (how tests/golden/trained_weights_lift_seed129.npz was made);
  * export_torch_state_dicts(path, trainer): torch.save of {net: state_dict} with rlkit's parameter names
    (fc0.weight ... last_fc_log_std.bias), loadable on a box that has rlkit with
    `policy.load_state_dict(torch.load(path)["policy"])`."""
from __future__ import annotations

import json
import os
import shutil
import zipfile
import zlib
from collections import OrderedDict

import numpy as np

FORMAT = "robosuite_benchmark_amd.checkpoint/1"     # (generations + CRC32 are additions the loader treats as optional)
NETS = ("policy", "qf1", "qf2", "target_qf1", "target_qf2")


def _save(dirname, manifest, name, arr):
    arr = np.ascontiguousarray(arr)
    path = os.path.join(dirname, name + ".npy")
    with open(path, "wb") as f:
        np.save(f, arr, allow_pickle=False)
        f.flush()
        os.fsync(f.fileno())
    manifest["arrays"][name] = dict(dtype=str(arr.dtype), shape=list(arr.shape),
                                    crc32=int(zlib.crc32(memoryview(arr).cast("B")) & 0xFFFFFFFF))


def _fsync_dir(path):
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
    finally:
        os.close(fd)


def _generations(dirname):
    return sorted(int(d[4:]) for d in os.listdir(dirname) if d.startswith("gen-") and d[4:].isdigit())


def current_dir(dirname):
    """The directory holding the newest COMPLETE save under `dirname`, or None when there is no save at all.  Saves are
    generations `<dirname>/gen-<n>/`; `<dirname>/latest` names the newest complete one and is replaced atomically, last.
    A `latest` that names a generation without a manifest (somebody removed or damaged it) is NOT "no checkpoint" -- a
    resuming run would silently start from scratch and its first save would delete what is left: the newest other
    generation that has a manifest is used (load_checkpoint verifies its CRCs), then the flat layout of the first format
    revision; only if neither exists the dangling pointer is an error."""
    if not os.path.isdir(dirname):
        return None
    ptr = os.path.join(dirname, "latest")
    flat = os.path.exists(os.path.join(dirname, "manifest.json"))
    if os.path.exists(ptr):
        with open(ptr) as f:
            sub = f.read().strip()
        cand = os.path.join(dirname, sub)
        if sub and os.path.exists(os.path.join(cand, "manifest.json")):
            return cand
        for g in reversed(_generations(dirname)):
            alt = os.path.join(dirname, f"gen-{g}")
            if os.path.exists(os.path.join(alt, "manifest.json")):
                return alt
        if flat:
            return dirname
        raise FileNotFoundError(f"{dirname}: `latest` names '{sub}', which holds no manifest, and no other complete "
                                "generation exists -- refusing to treat a damaged checkpoint directory as 'no checkpoint'")
    if flat:                                                          # flat layout of the first format revision
        return dirname
    return None


def checkpoint_exists(dirname):
    return current_dir(dirname) is not None


def save_checkpoint(dirname, trainer, replay_buffer=None, extra=None):
    """Write a new generation `<dirname>/gen-<n>/` (manifest.json + one .npy per array), then flip `<dirname>/latest`
    to it and drop the older generations.  Nothing of the previous save is touched before the pointer has flipped, so a
    process killed at ANY point leaves either the old or the new save complete -- never a mix (every array also carries
    a CRC32 in the manifest, checked on load).  `extra`: JSON-serialisable dict (epoch, seed ...)."""
    os.makedirs(dirname, exist_ok=True)
    if trainer._h is None:
        raise RuntimeError("the trainer owns no device state yet (construct it with batch_size= or train once)")
    gens = _generations(dirname)
    sub = f"gen-{(gens[-1] + 1) if gens else 0}"
    gdir = os.path.join(dirname, sub)
    os.makedirs(gdir)
    man = dict(format=FORMAT, arrays=OrderedDict(), extra=extra or {})
    st = trainer.state_dict()
    for net, flat in st["params"].items():
        _save(gdir, man, f"params.{net}", flat)
    for net, (m, v) in st["opt"].items():
        _save(gdir, man, f"adam_m.{net}", m)
        _save(gdir, man, f"adam_v.{net}", v)
    _save(gdir, man, "trainer_scalars", st["scalars"])      # log_alpha, its Adam m/v, adam_t, n_train_steps_total, alpha
    man["trainer"] = dict(obs_dim=trainer.obs_dim, action_dim=trainer.act_dim, num_train_steps=trainer._num_train_steps,
                          batch_size=trainer._batch,
                          hparams={k: getattr(trainer, k) for k in (
                              "discount", "reward_scale", "policy_lr", "qf_lr", "soft_target_tau", "target_update_period",
                              "use_automatic_entropy_tuning", "target_entropy", "noise_seed")})
    if replay_buffer is not None:
        bs = replay_buffer.state_dict()
        for k in ("observations", "actions", "rewards", "next_observations", "terminals", "rng_key"):
            _save(gdir, man, "buffer." + k, bs[k])
        man["buffer"] = {k: int(bs[k]) for k in ("capacity", "obs_dim", "action_dim", "top", "size", "rng_pos")}
    with open(os.path.join(gdir, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
        f.flush()
        os.fsync(f.fileno())
    _fsync_dir(gdir)
    tmp = os.path.join(dirname, "latest.tmp")
    with open(tmp, "w") as f:
        f.write(sub + "\n")
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, os.path.join(dirname, "latest"))          # the commit point
    _fsync_dir(dirname)
    for g in gens:                                             # older generations (and torn ones) go only now
        shutil.rmtree(os.path.join(dirname, f"gen-{g}"), ignore_errors=True)
    return man


def load_checkpoint(dirname, trainer, replay_buffer=None):
    """Restore trainer (and buffer) state saved by save_checkpoint; returns the manifest's `extra`."""
    cdir = current_dir(dirname)
    if cdir is None:
        raise FileNotFoundError(f"{dirname}: no complete checkpoint")
    with open(os.path.join(cdir, "manifest.json")) as f:
        man = json.load(f)
    if man.get("format") != FORMAT:
        raise ValueError(f"{cdir}: not a {FORMAT} checkpoint")

    def arr(name):
        a = np.load(os.path.join(cdir, name + ".npy"), allow_pickle=False)
        want = man["arrays"][name]
        if str(a.dtype) != want["dtype"] or list(a.shape) != want["shape"]:
            raise ValueError(f"{cdir}/{name}.npy does not match the manifest")
        if "crc32" in want and int(zlib.crc32(memoryview(np.ascontiguousarray(a)).cast("B")) & 0xFFFFFFFF) != want["crc32"]:
            raise ValueError(f"{cdir}/{name}.npy: content does not match the manifest (torn or foreign file)")
        return a

    tm = man["trainer"]
    if (tm["obs_dim"], tm["action_dim"]) != (trainer.obs_dim, trainer.act_dim):
        raise ValueError("checkpointed trainer has other dimensions")
    # read and verify EVERYTHING before touching the live state
    st = dict(params={n: arr(f"params.{n}") for n in trainer.NETS},
              opt={n: (arr(f"adam_m.{n}"), arr(f"adam_v.{n}")) for n in ("policy", "qf1", "qf2")},
              scalars=arr("trainer_scalars"))
    bs = None
    if replay_buffer is not None:
        if "buffer" not in man:
            raise ValueError(f"{cdir} holds no replay buffer")
        bs = dict(man["buffer"])
        for k in ("observations", "actions", "rewards", "next_observations", "terminals", "rng_key"):
            bs[k] = arr("buffer." + k)
    if trainer._h is None:
        trainer._create(int(tm["batch_size"]))
    trainer.load_state_dict(st)
    trainer._num_train_steps = int(tm["num_train_steps"])
    if bs is not None:
        replay_buffer.load_state_dict(bs)
    return man.get("extra", {})


# ---- the reference's own checkpoint files ----------------------------------------------------------------
def rlkit_layer_shapes(obs_dim, action_dim, hidden=(256, 256), agent="SAC", hidden_q=None):
    """(name, shape) of every parameter in rlkit's registration order (Mlp: fc0.., last_fc; the SAC policy adds
    last_fc_log_std, the TD3 TanhMlpPolicy does not) -- the order torch.save numbers the storages in."""
    def mlp(inp, outs, hidden=hidden):
        names, k = [], inp
        for i, h in enumerate(hidden):
            names += [(f"fc{i}.weight", (h, k)), (f"fc{i}.bias", (h,))]
            k = h
        for head, n in outs:
            names += [(f"{head}.weight", (n, k)), (f"{head}.bias", (n,))]
        return names
    heads = [("last_fc", action_dim)] + ([("last_fc_log_std", action_dim)] if agent == "SAC" else [])
    hq = tuple(hidden_q) if hidden_q else tuple(hidden)
    return OrderedDict(policy=mlp(obs_dim, heads),
                       qf1=mlp(obs_dim + action_dim, [("last_fc", 1)], hq),
                       qf2=mlp(obs_dim + action_dim, [("last_fc", 1)], hq))


def read_rlkit_zip_params(path, obs_dim, action_dim, hidden=(256, 256)):
    """Flat nn.Linear vectors {policy, qf1, qf2} from the raw storages of a ZIP-format params.pkl.
    Uses zipfile only -- nothing in the file is unpickled or executed.  The shipped snapshots alias
    target_qf* to qf* (20 storages), so three nets come back.  Raises on a legacy (non-zip) file."""
    if not zipfile.is_zipfile(path):
        raise ValueError(f"{path}: legacy (pre-zip) torch checkpoint -- cannot be read without unpickling")
    z = zipfile.ZipFile(path)
    names = [n for n in z.namelist() if "/data/" in n and not n.endswith("/")]
    root = names[0].split("/data/")[0]
    out, k = OrderedDict(), 0
    for net, layers in rlkit_layer_shapes(obs_dim, action_dim, hidden).items():
        parts = []
        for pname, shape in layers:
            a = np.frombuffer(z.read(f"{root}/data/{k}"), dtype="<f4")
            if a.size != int(np.prod(shape)):
                raise ValueError(f"{path}: storage {k} has {a.size} floats, {net}.{pname} needs {shape}")
            parts.append(a)
            k += 1
        out[net] = np.concatenate(parts).astype(np.float32)
    return out


def export_torch_state_dicts(path, source, obs_dim=None, action_dim=None):
    """torch.save({net: OrderedDict(rlkit parameter name -> tensor)}) for the networks of `source`: a SACTrainer,
    or a dict {net: flat nn.Linear vector} together with obs_dim / action_dim."""
    import torch                      # only the exporter needs torch; the library itself does not
    hidden, hidden_q = (256, 256), None
    if isinstance(source, dict):
        st = source
    else:
        st, obs_dim, action_dim = source.state_dict()["params"], source.obs_dim, source.act_dim
        hidden, hidden_q = tuple(source.policy.hidden_sizes), tuple(source.qf1.hidden_sizes)
    shapes = rlkit_layer_shapes(obs_dim, action_dim, hidden, agent="TD3" if "target_policy" in st else "SAC", hidden_q=hidden_q)
    shapes["target_qf1"], shapes["target_qf2"], shapes["target_policy"] = shapes["qf1"], shapes["qf2"], shapes["policy"]
    out = OrderedDict()
    for net in NETS + ("target_policy",):
        if net not in st:
            continue
        flat, off, sd = np.asarray(st[net], np.float32), 0, OrderedDict()
        for pname, shape in shapes[net]:
            n = int(np.prod(shape))
            sd[pname] = torch.from_numpy(flat[off:off + n].reshape(shape).copy())
            off += n
        if off != flat.size:
            raise ValueError(f"{net}: {flat.size} parameters, layout needs {off}")
        out[net] = sd
    torch.save(out, path)
    return out
