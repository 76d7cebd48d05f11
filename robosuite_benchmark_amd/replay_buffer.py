"""EnvReplayBuffer living in HBM (mirror of rlkit's EnvReplayBuffer / SimpleReplayBuffer).

Reference call sites: /root/reference/util/rlkit_utils.py:139-142 (constructor
``EnvReplayBuffer(replay_buffer_size, expl_env)``), /root/reference/util/rlkit_custom.py:207,230
(``add_paths``), :235-236 (``random_batch``), :62,80 (snapshot), :250-253 (``get_diagnostics``)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib


def _space_dim(space):
    if hasattr(space, "low"):
        return int(np.asarray(space.low).size)
    return int(space)


def numpy_global_state_address():
    """Address of the state struct ``{uint32_t key[624]; int pos;}`` behind ``np.random`` -- the legacy global
    ``RandomState``'s MT19937 bit generator, through the ctypes interface NumPy documents for bit generators
    (``BitGenerator.ctypes.state_address``).  ``np.random.seed`` / ``set_state`` / every draw update it in place.  The
    layout is checked against ``np.random.get_state()``; None if this NumPy does not look like that."""
    try:
        bg = np.random.mtrand._rand._bit_generator
        addr = int(bg.ctypes.state_address)
        key = np.ctypeslib.as_array((C.c_uint32 * 624).from_address(addr))
        pos = C.c_int32.from_address(addr + 4 * 624).value
        st = np.random.get_state()
        if st[0] == "MT19937" and int(st[2]) == pos and np.array_equal(key, st[1]):
            return addr
    except Exception:
        pass
    return None


class DeviceBatch(dict):
    """The dict random_batch() returns, with its arrays still in HBM.  Any read access (batch["rewards"], iteration,
    len ...) first copies the five arrays to the host, after which it is an ordinary dict; SACTrainer.train()
    recognises an untouched one and trains on the device copy.  The device copy lives until 16 more batches have
    been drawn from the buffer; reading an older, never-read batch raises RuntimeError."""
    KEYS = ("observations", "actions", "rewards", "terminals", "next_observations")

    def __init__(self, buffer, token, batch_size):
        super().__init__()
        self._buffer, self._token, self._batch_size, self._on_host = buffer, token, batch_size, False

    @property
    def on_device(self):
        return not self._on_host

    def _fetch(self):
        if self._on_host:
            return
        buf, B = self._buffer, self._batch_size
        O, A = buf._observation_dim, buf._action_dim
        obs, nobs = np.empty((B, O), np.float32), np.empty((B, O), np.float32)
        act = np.empty((B, A), np.float32)
        rew, term = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        _lib.check(buf._lib.sac_read_batch_device(buf._h, self._token, _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew),
                                                  _lib.ptr(term), _lib.ptr(nobs), None), "sac_read_batch_device")
        self._on_host = True
        super().update(observations=obs, actions=act, rewards=rew, terminals=term, next_observations=nobs)

    def indices(self):
        """The sampled buffer indices of this batch (while it is still on the device)."""
        idx = np.empty(self._batch_size, np.int64)
        buf = self._buffer
        _lib.check(buf._lib.sac_read_batch_device(buf._h, self._token, None, None, None, None, None, _lib.ptr(idx)),
                   "sac_read_batch_device")
        return idx


def _fetching(name):
    def method(self, *a, **k):
        self._fetch()
        return getattr(dict, name)(self, *a, **k)
    method.__name__ = name
    return method


for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__repr__", "__eq__", "__ne__", "__setitem__", "__delitem__",
           "keys", "values", "items", "get", "copy", "pop", "update", "setdefault"):
    setattr(DeviceBatch, _n, _fetching(_n))


class EnvReplayBuffer:
    """``EnvReplayBuffer(max_replay_buffer_size, env)``; ``env`` only supplies
    observation_space / action_space sizes (pass ``obs_dim=`` / ``action_dim=`` instead when
    there is no env object).  Like rlkit's buffer, sampling consumes the process-wide NumPy legacy stream -- the one
    ``np.random.seed(args.seed)`` seeds (scripts/train.py:112): constructed the reference's way,
    ``EnvReplayBuffer(variant['replay_buffer_size'], expl_env)``, every ``random_batch`` draws exactly the indices
    ``np.random.randint(0, size, batch_size)`` would have and leaves ``np.random`` where that call would have left it,
    whatever else consumes ``np.random`` in between.  ``seed(int)`` / ``seed_from_numpy(rs)`` give the buffer a private
    stream instead (tests, several independent runs in one process)."""

    def __init__(self, max_replay_buffer_size, env=None, env_info_sizes=None, obs_dim=None, action_dim=None,
                 device=0, numpy_global_stream=True, lazy_batches=True):
        if env is not None:
            obs_dim = _space_dim(env.observation_space)
            action_dim = _space_dim(env.action_space)
        if obs_dim is None or action_dim is None:
            raise ValueError("EnvReplayBuffer needs an env or obs_dim/action_dim")
        self.env = env
        self._observation_dim, self._action_dim = int(obs_dim), int(action_dim)
        self._max_replay_buffer_size = int(max_replay_buffer_size)
        # True (default): every random_batch() continues the process-wide np.random stream -- exactly what rlkit's
        # np.random.randint call does, so host consumers of np.random (env resets, exploration noise) interleave
        # identically.  The library is BOUND to np.random's own state words (sac_rng_bind_host): each device draw is
        # mirrored on them in place, so there is no per-call state transfer and the read-ahead stays (the stepwise loop
        # runs at full speed); if this NumPy hides its state the fall-back is get_state / set_state around each call.
        self.numpy_global_stream = bool(numpy_global_stream)
        self.lazy_batches = bool(lazy_batches)     # random_batch() returns device-resident DeviceBatch dicts
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.sac_buffer_create(C.byref(h), self._max_replay_buffer_size, self._observation_dim,
                                               self._action_dim, int(device)), "sac_buffer_create")
        self._h = h
        self._bound = False
        if self.numpy_global_stream:
            self.bind_numpy_global_stream()

    def bind_numpy_global_stream(self):
        """(Re)join the process-wide ``np.random`` stream -- the construction default; undoes ``seed(int)``."""
        self.numpy_global_stream = True
        addr = numpy_global_state_address()
        if addr is not None:
            _lib.check(self._lib.sac_rng_bind_host(self._h, C.c_void_p(addr), C.c_void_p(addr + 4 * 624)),
                       "sac_rng_bind_host")
            self._bound = True

    def _unbind(self):
        self.numpy_global_stream = False
        if self._bound:
            _lib.check(self._lib.sac_rng_bind_host(self._h, None, None), "sac_rng_bind_host")
            self._bound = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sac_buffer_destroy(h)

    # ---- the np.random global stream ----------------------------------------------------
    def seed(self, seed: int):
        """A PRIVATE stream for this buffer, seeded like ``np.random.RandomState(seed)`` (it leaves the process-wide
        stream: np.random is neither read nor written from here on -- ``bind_numpy_global_stream()`` rejoins it)."""
        self._unbind()
        _lib.check(self._lib.sac_rng_seed(self._h, int(seed) & 0xFFFFFFFF), "sac_rng_seed")

    def seed_from_numpy(self, rs=None):
        """Adopt the state of a RandomState as a private stream; without an argument: (re)join np.random itself --
        a no-op on a buffer that never left it."""
        if rs is None or rs is np.random:
            if self._bound:
                return
            st = np.random.get_state()
        else:
            self._unbind()
            st = rs.get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        _lib.check(self._lib.sac_rng_set_state(self._h, _lib.ptr(key), int(st[2])), "sac_rng_set_state")

    def sync_to_numpy(self, rs=None):
        """Write the stream state back into np.random / a RandomState (a buffer bound to np.random keeps it right by
        itself: nothing to do)."""
        if self._bound and (rs is None or rs is np.random):
            return
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        _lib.check(self._lib.sac_rng_get_state(self._h, _lib.ptr(key), C.byref(pos)), "sac_rng_get_state")
        (rs or np.random).set_state(("MT19937", key, int(pos.value), 0, 0.0))

    def rng_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        _lib.check(self._lib.sac_rng_get_state(self._h, _lib.ptr(key), C.byref(pos)), "sac_rng_get_state")
        return key, int(pos.value)

    # ---- rlkit ReplayBuffer interface ----------------------------------------------------
    def add_sample(self, observation, action, reward, next_observation, terminal, **kwargs):
        self.add_block(np.asarray(observation)[None], np.asarray(action)[None], np.asarray([reward]),
                       np.asarray(next_observation)[None], np.asarray([terminal]))

    def add_block(self, observations, actions, rewards, next_observations, terminals):
        n = len(observations)
        t = np.ascontiguousarray(np.asarray(terminals).reshape(n) != 0, dtype=np.uint8)
        o, a, no = (np.asarray(x) for x in (observations, actions, next_observations))
        r = np.asarray(rewards).reshape(n)
        if o.dtype == np.float32 and a.dtype == np.float32 and no.dtype == np.float32:
            o, a, r, no = map(_lib.f32, (o, a, r, no))
            fn = self._lib.sac_buffer_add
        else:   # the reference's native float64 paths
            o, a, r, no = (np.ascontiguousarray(x, dtype=np.float64) for x in (o, a, r, no))
            fn = self._lib.sac_buffer_add_f64
        assert o.shape == (n, self._observation_dim) and a.shape == (n, self._action_dim)
        _lib.check(fn(self._h, n, _lib.ptr(o), _lib.ptr(a), _lib.ptr(r), _lib.ptr(no), _lib.ptr(t)),
                   "sac_buffer_add")

    def ingest_pending(self):
        """True while inserted rows are still on their way to HBM (inserts are asynchronous: add_* returns once the
        rows sit in pinned staging and their copies are enqueued; sampling is ordered behind them on the device)."""
        return bool(_lib.check(self._lib.sac_buffer_ingest_pending(self._h), "sac_buffer_ingest_pending"))

    def ingest_wait(self):
        _lib.check(self._lib.sac_buffer_ingest_wait(self._h), "sac_buffer_ingest_wait")

    def add_path(self, path):
        self.add_block(path["observations"], path["actions"], path["rewards"], path["next_observations"],
                       path["terminals"])
        self.terminate_episode()

    def add_paths(self, paths):
        for path in paths:
            self.add_path(path)

    def terminate_episode(self):
        pass

    def num_steps_can_sample(self):
        return int(self._lib.sac_buffer_size(self._h))

    # ---- checkpointing (the reference saves no buffer: rlkit get_snapshot() is {}) -----------------
    def read_rows(self, start, n):
        """Storage rows [start, start+n) as (obs, act, rew, next_obs, term) in the add_block layout."""
        n = int(n)
        O, A = self._observation_dim, self._action_dim
        o, no = np.empty((n, O), np.float32), np.empty((n, O), np.float32)
        a, r, t = np.empty((n, A), np.float32), np.empty((n, 1), np.float32), np.empty((n, 1), np.uint8)
        _lib.check(self._lib.sac_buffer_read(self._h, int(start), n, *(map(_lib.ptr, (o, a, r, no, t)))),
                   "sac_buffer_read")
        return o, a, r, no, t

    def state_dict(self):
        """Everything a bit-exact resume needs: the valid rows in storage order, ring cursor, generator state."""
        size, top = int(self._lib.sac_buffer_size(self._h)), int(self._lib.sac_buffer_top(self._h))
        o, a, r, no, t = self.read_rows(0, size)                  # the ring fills from row 0
        key, pos = self.rng_state()
        return dict(capacity=self._max_replay_buffer_size, obs_dim=self._observation_dim, action_dim=self._action_dim,
                    top=top, size=size, observations=o, actions=a, rewards=r, next_observations=no, terminals=t,
                    rng_key=key, rng_pos=pos)

    def load_state_dict(self, st, chunk=1 << 18):
        if (int(st["capacity"]), int(st["obs_dim"]), int(st["action_dim"])) != (
                self._max_replay_buffer_size, self._observation_dim, self._action_dim):
            raise ValueError("checkpointed replay buffer has another shape")
        size = int(st["size"])
        _lib.check(self._lib.sac_buffer_set_cursor(self._h, 0, 0), "sac_buffer_set_cursor")
        for i in range(0, size, chunk):
            j = min(size, i + chunk)
            self.add_block(*(np.asarray(st[k][i:j]) for k in ("observations", "actions", "rewards",
                                                               "next_observations", "terminals")))
        _lib.check(self._lib.sac_buffer_set_cursor(self._h, int(st["top"]), size), "sac_buffer_set_cursor")
        key = np.ascontiguousarray(st["rng_key"], dtype=np.uint32)
        _lib.check(self._lib.sac_rng_set_state(self._h, _lib.ptr(key), int(st["rng_pos"])), "sac_rng_set_state")

    def random_batch(self, batch_size, return_indices=False, lazy=None):
        """rlkit's random_batch.  By default (lazy) the batch is drawn and gathered on the device and STAYS there:
        the returned dict copies itself to the host only when somebody reads it, and SACTrainer.train() takes it
        straight from HBM -- the reference's unmodified loop (rlkit_custom.py:235-238) then runs without a PCIe
        round trip or a synchronisation per step.  lazy=False / return_indices=True: plain dict of host arrays."""
        if lazy is None:
            lazy = self.lazy_batches and not return_indices
        slow_global = self.numpy_global_stream and not self._bound      # (a NumPy whose state words are out of reach)
        if lazy:
            if slow_global:
                self.seed_from_numpy()
            tok = C.c_int64()
            _lib.check(self._lib.sac_random_batch_device(self._h, int(batch_size), C.byref(tok)),
                       "sac_random_batch_device")
            if slow_global:
                self.sync_to_numpy()
            return DeviceBatch(self, int(tok.value), int(batch_size))
        B, O, A = int(batch_size), self._observation_dim, self._action_dim
        obs, nobs = np.empty((B, O), np.float32), np.empty((B, O), np.float32)
        act = np.empty((B, A), np.float32)
        rew, term = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        idx = np.empty(B, np.int64)
        if slow_global:
            self.seed_from_numpy()
        _lib.check(self._lib.sac_random_batch(self._h, B, _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew),
                                              _lib.ptr(term), _lib.ptr(nobs), _lib.ptr(idx)), "sac_random_batch")
        if slow_global:
            self.sync_to_numpy()
        batch = dict(observations=obs, actions=act, rewards=rew, terminals=term, next_observations=nobs)
        return (batch, idx) if return_indices else batch

    def gather(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        B, O, A = len(idx), self._observation_dim, self._action_dim
        obs, nobs = np.empty((B, O), np.float32), np.empty((B, O), np.float32)
        act = np.empty((B, A), np.float32)
        rew, term = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        _lib.check(self._lib.sac_gather(self._h, _lib.ptr(idx), B, _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew),
                                        _lib.ptr(term), _lib.ptr(nobs)), "sac_gather")
        return dict(observations=obs, actions=act, rewards=rew, terminals=term, next_observations=nobs)

    def sample_indices(self, batch_size, n_batches=1):
        idx = np.empty(int(batch_size) * int(n_batches), np.int64)
        _lib.check(self._lib.sac_sample_indices(self._h, int(batch_size), int(n_batches), _lib.ptr(idx)),
                   "sac_sample_indices")
        return idx.reshape(int(n_batches), int(batch_size))

    def sample_gather_device(self, batch_size, n_batches):
        ms = np.zeros(2, np.float32)
        _lib.check(self._lib.sac_sample_gather_device(self._h, int(batch_size), int(n_batches), _lib.ptr(ms)),
                   "sac_sample_gather_device")
        return float(ms[0]), float(ms[1])

    def read_slot(self, slot, batch_size):
        B, O, A = int(batch_size), self._observation_dim, self._action_dim
        obs, nobs = np.empty((B, O), np.float32), np.empty((B, O), np.float32)
        act = np.empty((B, A), np.float32)
        rew, term = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        idx = np.empty(B, np.int64)
        _lib.check(self._lib.sac_read_slot(self._h, int(slot), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew),
                                           _lib.ptr(term), _lib.ptr(nobs), _lib.ptr(idx)), "sac_read_slot")
        return dict(observations=obs, actions=act, rewards=rew, terminals=term, next_observations=nobs), idx

    def get_diagnostics(self):
        return OrderedDict([("size", self.num_steps_can_sample())])

    def get_snapshot(self):
        return {}

    def end_epoch(self, epoch):
        return
