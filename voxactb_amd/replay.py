"""Replay store and batch stream of the training loop
(reference: YARR/yarr/replay_buffer/uniform_replay_buffer.py:322-386 add / add_final, :639-756 sample_transition_batch;
task_uniform_replay_buffer.py:30-64 per-task index lists, :66-131 task-uniform sampling with the DDP stride
`task_idxs[task][rank:total:num_replicas]` (:103-108); wrappers/pytorch_replay_buffer.py:58-82 the DataLoader wrapper).

What the reference does: one pickle per transition on disk (or object arrays in RAM), unpickled on EVERY sample, batch
assembled element by element in Python, handed to a DataLoader with `pin_memory=True`, copied to the device by the runner
(`v.to(device)` per key, offline_train_runner.py:140) -- "Sample time" becomes the bottleneck once a step takes 0.18 s.

Here:
  * `ShardReplayBuffer` keeps every element as a COLUMN: fixed-size binary shards `[rows_per_shard, *shape]` (numpy arrays, or
    `np.memmap` files `<save_dir>/<element>.<shard>.bin` when a directory is given -- no pickle anywhere), so a batch is a
    fancy-indexed gather per element straight into a staging buffer.  Same `add` / `add_final` / `sample_transition_batch`
    semantics for the configuration VoxAct-B uses (timesteps = 1, update_horizon = 1): a transition's `_tp1` observations are
    the next row's, terminal rows are followed by the episode's final observation (terminal = -1 marks it unsampleable),
    batches are task-uniform and every rank draws from its own stride of each task's rows.
  * `DeviceBatchStream` samples ahead on a background thread into a ring of PINNED staging buffers and copies each batch to the
    GPU on its own HIP stream; the iterator hands out device tensors guarded by an event, so the next batch's host gather and
    H2D copy overlap with the current training step and the runner's `.to(device)` is a no-op.
"""
import collections
import math
import os
import threading

import numpy as np

ACTION, REWARD, TERMINAL, TIMEOUT, INDICES, TASK = 'action', 'reward', 'terminal', 'timeout', 'indices', 'task'


class ReplayElement:
    def __init__(self, name, shape, type, is_observation=False):
        self.name, self.shape, self.type, self.is_observation = name, tuple(shape), type, is_observation


class ObservationElement(ReplayElement):
    def __init__(self, name, shape, type):
        super().__init__(name, shape, type, True)


def _is_numeric(dtype):
    return dtype not in (str, object) and np.dtype(dtype).kind in 'biuf'


class _Column:
    """One replay element: rows in shards of `rows` transitions, in RAM or memory-mapped files."""

    def __init__(self, name, shape, dtype, rows, save_dir):
        self.name, self.shape, self.rows, self.save_dir = name, tuple(shape), rows, save_dir
        self.numeric = _is_numeric(dtype)
        self.dtype = np.dtype(dtype) if self.numeric else object
        self.shards = []

    def _shard(self, k):
        while len(self.shards) <= k:
            if self.numeric and self.save_dir is not None:
                path = os.path.join(self.save_dir, '%s.%05d.bin' % (self.name, len(self.shards)))
                self.shards.append(np.memmap(path, dtype=self.dtype, mode='w+', shape=(self.rows,) + self.shape))
            elif self.numeric:
                self.shards.append(np.zeros((self.rows,) + self.shape, dtype=self.dtype))
            else:
                self.shards.append(np.empty((self.rows,) + self.shape, dtype=object))
        return self.shards[k]

    def write(self, row, value):
        sh = self._shard(row // self.rows)
        if self.numeric:
            sh[row % self.rows] = np.asarray(value, dtype=self.dtype).reshape(self.shape)
        elif self.shape:
            sh[row % self.rows] = np.asarray(value, dtype=object).reshape(self.shape)
        else:
            sh[row % self.rows] = value

    def gather(self, rows, out):
        """out[i] = column[rows[i]]; rows grouped by shard so that each shard is indexed once."""
        rows = np.asarray(rows)
        which = rows // self.rows
        for k in np.unique(which):
            sel = np.nonzero(which == k)[0]
            out[sel] = self.shards[int(k)][rows[sel] % self.rows]
        return out


class ShardReplayBuffer:
    """Drop-in for yarr's TaskUniformReplayBuffer as launch_utils.create_replay configures it (launch_utils.py:148-163)."""

    def __init__(self, batch_size=32, timesteps=1, replay_capacity=int(1e6), update_horizon=1, gamma=0.99,
                 max_sample_attempts=10000, action_shape=(), action_dtype=np.float32, reward_shape=(), reward_dtype=np.float32,
                 observation_elements=None, extra_replay_elements=None, save_dir=None, purge_replay_on_shutdown=True,
                 num_replicas=None, rank=None, rows_per_shard=256):
        if timesteps != 1 or update_horizon != 1:
            raise NotImplementedError('PerAct / VoxAct-B train with timesteps = 1, update_horizon = 1 (conf/config.yaml:40-41, '
                                      'launch_utils.py:158); frame stacking and n-step returns are not built')
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            ok = dist.is_available() and dist.is_initialized()
            num_replicas = dist.get_world_size() if (num_replicas is None and ok) else (num_replicas or 1)
            rank = dist.get_rank() if (rank is None and ok) else (rank or 0)
        if not 0 <= rank < num_replicas:
            raise ValueError('Invalid rank %d, rank should be in the interval [0, %d]' % (rank, num_replicas - 1))
        self._batch_size, self._timesteps, self._replay_capacity = batch_size, timesteps, int(replay_capacity)
        self._update_horizon, self._gamma, self._max_sample_attempts = update_horizon, gamma, max_sample_attempts
        self._action_shape, self._action_dtype = tuple(action_shape), action_dtype
        self._reward_shape, self._reward_dtype = tuple(reward_shape), reward_dtype
        self._observation_elements = list(observation_elements or [])
        self._extra_replay_elements = list(extra_replay_elements or [])
        self._rank, self._num_replicas = rank, num_replicas
        self._save_dir, self._purge = save_dir, purge_replay_on_shutdown
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
        self._storage_signature = [ReplayElement(ACTION, self._action_shape, action_dtype),
                                   ReplayElement(REWARD, self._reward_shape, reward_dtype),
                                   ReplayElement(TERMINAL, (), np.int8), ReplayElement(TIMEOUT, (), bool)]
        # (upstream passes plain ReplayElements inside `observation_elements` too -- the discrete actions, pose, language:
        # whatever is in that list is stored per row, required by add_final and sampled with a `_tp1` twin)
        self._obs_signature = [ReplayElement(e.name, e.shape, e.type, True) for e in self._observation_elements]
        self._storage_signature += self._obs_signature + self._extra_replay_elements
        self._cols = {e.name: _Column(e.name, e.shape, e.type, rows_per_shard, save_dir) for e in self._storage_signature}
        self._task_idxs = collections.OrderedDict()
        self._add_count = 0
        self._lock = threading.Lock()
        self._rng = np.random.default_rng()

    # ------------------------------------------------------------------ properties the runner / launch code reads
    batch_size = property(lambda self: self._batch_size)
    timesteps = property(lambda self: self._timesteps)
    replay_capacity = property(lambda self: self._replay_capacity)
    add_count = property(lambda self: np.array(self._add_count))

    def is_empty(self):
        return self._add_count == 0

    def is_full(self):
        return self._add_count >= self._replay_capacity

    def cursor(self):
        return self._add_count % self._replay_capacity

    def seed(self, s):
        self._rng = np.random.default_rng(s)

    # ------------------------------------------------------------------ writing (uniform_replay_buffer.py:322-386)
    def _check(self, kwargs, signature):
        if len(kwargs) != len(signature):
            raise ValueError('Add expects %d elements, received %d.\nList of expected:\n%s\nList of actual:\n%s' % (
                len(signature), len(kwargs), sorted(e.name for e in signature), sorted(kwargs)))
        for e in signature:
            v = kwargs[e.name]
            shape = v.shape if isinstance(v, np.ndarray) else (np.array(v).shape if isinstance(v, (tuple, list)) else ())
            if tuple(shape) != tuple(e.shape):
                raise ValueError('arg has shape %s, expected %s' % (tuple(shape), tuple(e.shape)))

    def _write(self, row_values, task):
        with self._lock:
            if self.is_full():
                raise RuntimeError('ShardReplayBuffer is full (capacity %d): the offline demo replay never wraps around'
                                   % self._replay_capacity)
            row = self._add_count
            for name, v in row_values.items():
                self._cols[name].write(row, v)
            if task is not None:
                self._task_idxs.setdefault(task, []).append(row)
            self._add_count += 1

    def add(self, action, reward, terminal, timeout, **kwargs):
        kwargs[ACTION], kwargs[REWARD], kwargs[TERMINAL], kwargs[TIMEOUT] = action, reward, terminal, timeout
        self._check(kwargs, self._storage_signature)
        self._write(kwargs, kwargs[TASK] if TASK in kwargs else '')

    def add_final(self, **kwargs):
        """The observation after an episode's last transition: stored as a row of its own with terminal = -1 (never sampled,
        only read as the `_tp1` of the row before it)."""
        self._check(kwargs, self._obs_signature)
        row = {}
        for e in self._storage_signature:
            if e.name in kwargs:
                row[e.name] = kwargs[e.name]
            elif e.name == TERMINAL:
                row[e.name] = -1
            else:
                row[e.name] = np.zeros(e.shape, e.type) if _is_numeric(e.type) else None
        self._write(row, kwargs.get(TASK, ''))

    # ------------------------------------------------------------------ sampling (task_uniform_replay_buffer.py:66-131)
    def _is_valid(self, row):
        return 0 <= row < self._add_count - self._update_horizon and int(self._cols[TERMINAL].shards[row // self._cols[TERMINAL].rows][
            row % self._cols[TERMINAL].rows]) != -1

    def sample_index_batch(self, batch_size):
        if self._add_count - self._update_horizon <= 0:
            raise RuntimeError('Cannot sample a batch with fewer than stack size (%d) + update_horizon (%d) transitions.'
                               % (self._timesteps, self._update_horizon))
        tasks = list(self._task_idxs.keys())
        picked = self._rng.choice(len(tasks), batch_size, replace=batch_size > len(tasks))
        out = []
        for t in picked:
            rows = self._task_idxs[tasks[int(t)]]
            total = math.ceil(len(rows) / self._num_replicas) * self._num_replicas
            mine = rows[self._rank:total:self._num_replicas]          # the DDP stride: every rank sees its own fraction
            for _ in range(self._max_sample_attempts):
                r = mine[int(self._rng.integers(len(mine)))] if mine else -1
                if self._is_valid(r):
                    out.append(r)
                    break
            else:
                raise RuntimeError('Max sample attempts: Tried %d times but only sampled %d valid indices. Batch size is %d'
                                   % (self._max_sample_attempts, len(out), batch_size))
        return out

    def get_transition_elements(self, batch_size=None):
        b = self._batch_size if batch_size is None else batch_size
        T = self._timesteps
        el = [ReplayElement(ACTION, (b, T) + self._action_shape, self._action_dtype),
              ReplayElement(REWARD, (b, T) + self._reward_shape, self._reward_dtype),
              ReplayElement(TERMINAL, (b, T), np.int8), ReplayElement(TIMEOUT, (b, T), bool), ReplayElement(INDICES, (b, T), np.int32)]
        for e in self._observation_elements:
            el.append(ReplayElement(e.name, (b, T) + tuple(e.shape), e.type, True))
            el.append(ReplayElement(e.name + '_tp1', (b, T) + tuple(e.shape), e.type, True))
        for e in self._extra_replay_elements:
            el.append(ReplayElement(e.name, (b,) + tuple(e.shape), e.type))
        return el

    def sample_transition_batch(self, batch_size=None, indices=None, pack_in_dict=True, out=None):
        """-> OrderedDict name -> array, shapes as get_transition_elements(); `task` is dropped as upstream does
        (uniform_replay_buffer.py:750-754).  `out`: optional dict of preallocated (pinned) arrays to fill."""
        b = self._batch_size if batch_size is None else batch_size
        with self._lock:
            rows = np.asarray(self.sample_index_batch(b) if indices is None else indices)
            if len(rows) != b:
                raise ValueError('need %d indices' % b)
            nxt = rows + 1
            batch = collections.OrderedDict()
            for e in self.get_transition_elements(b):
                if e.name in (TASK, TASK + '_tp1'):
                    continue
                numeric = _is_numeric(e.type)
                if out is not None and e.name in out:
                    arr = out[e.name]
                elif numeric:
                    arr = np.empty(e.shape, dtype=e.type)
                else:
                    arr = np.empty(e.shape, dtype=object)
                if e.name == INDICES:
                    arr[:, 0] = rows
                elif e.name == TERMINAL:
                    term = self._cols[TERMINAL].gather(rows, np.empty(b, np.int8))
                    arr[:, 0] = term != 0
                elif e.is_observation:
                    src, which = (e.name[:-4], nxt) if e.name.endswith('_tp1') else (e.name, rows)
                    self._cols[src].gather(which, arr[:, 0])
                elif e.name in (ACTION, REWARD, TIMEOUT):
                    self._cols[e.name].gather(rows, arr[:, 0])
                else:
                    self._cols[e.name].gather(rows, arr)
                batch[e.name] = arr
        return batch if pack_in_dict else tuple(batch.values())

    def shutdown(self):
        if self._save_dir is not None and self._purge:
            for col in self._cols.values():
                for sh in col.shards:
                    if isinstance(sh, np.memmap):
                        path = sh.filename
                        del sh
                        try:
                            os.remove(path)
                        except OSError:
                            pass
                col.shards = []


class DeviceBatchStream:
    """Iterator of device-resident batches: a producer thread gathers the next batches into pinned staging buffers and copies
    them to the GPU on a side stream while the consumer trains on the current one.

        stream = DeviceBatchStream(replay, device=rank, depth=2)
        for batch in stream: agent.update(i, batch)            # batch: dict name -> torch tensor on `device`

    Non-numeric elements (`lang_goal` strings) are dropped, as the runner drops them (offline_train_runner.py:140)."""

    def __init__(self, replay, device, depth=2, batch_size=None):
        import torch
        self._torch = torch
        self._replay = replay
        self._dev = torch.device('cuda:%d' % device) if isinstance(device, int) else torch.device(device)
        if self._dev.type != 'cuda':
            raise ValueError('DeviceBatchStream feeds a HIP device')
        self._b = batch_size or replay.batch_size
        self._depth = max(2, int(depth))
        self._numeric = [e for e in replay.get_transition_elements(self._b) if _is_numeric(e.type) and not e.name.startswith(TASK)]
        # ring of pinned staging buffers + their device twins
        self._host = [{e.name: torch.empty(e.shape, dtype=torch.from_numpy(np.empty(0, e.type)).dtype).pin_memory()
                       for e in self._numeric} for _ in range(self._depth)]
        self._devb = [{k: torch.empty_like(v, device=self._dev) for k, v in h.items()} for h in self._host]
        self._copy_stream = torch.cuda.Stream(device=self._dev)
        self._ready = [None] * self._depth              # event: H2D of slot i finished
        self._consumed = [None] * self._depth           # event: the training step that used slot i was enqueued
        self._free = threading.Semaphore(self._depth)
        self._full = threading.Semaphore(0)
        self._stop = False
        self._error = None
        self._head = 0
        self._last = None
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    def _produce(self):
        torch = self._torch
        slot = 0
        try:
            torch.cuda.set_device(self._dev)
            while not self._stop:
                self._free.acquire()
                if self._stop:
                    break
                host = self._host[slot]
                if self._ready[slot] is not None:
                    # the previous copy OUT of this pinned buffer may still be queued (it waits on the GPU for the step that
                    # read the slot's device twin, and the consumer frees a slot when that step is ENQUEUED, not executed):
                    # refilling the buffer before it ran would hand the GPU torn or later data.  Host-side wait, producer
                    # thread only.
                    self._ready[slot].synchronize()
                views = {k: v.numpy() for k, v in host.items()}
                self._replay.sample_transition_batch(self._b, out=views)
                with torch.cuda.stream(self._copy_stream):
                    if self._consumed[slot] is not None:
                        self._copy_stream.wait_event(self._consumed[slot])      # the step that read this slot has been issued
                    for k, v in host.items():
                        self._devb[slot][k].copy_(v, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                self._ready[slot] = ev
                self._full.release()
                slot = (slot + 1) % self._depth
        except Exception as e:  # noqa: BLE001
            self._error = e
            self._full.release()

    def __iter__(self):
        return self

    def __next__(self):
        """The batch handed out by the PREVIOUS call is released here: whatever the consumer did with it is on its stream by
        now, so an event recorded at this point lets the copy stream refill that slot safely."""
        torch = self._torch
        cur = torch.cuda.current_stream(self._dev)
        if self._last is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._consumed[self._last] = ev
            self._last = None
            self._free.release()
        self._full.acquire()
        if self._error is not None:
            raise self._error
        slot = self._head
        self._head = (slot + 1) % self._depth
        cur.wait_event(self._ready[slot])
        self._last = slot
        return dict(self._devb[slot])

    def close(self):
        self._stop = True
        self._free.release()
        self._thread.join(timeout=5)


class BatchStreamReplayBuffer:
    """Stands where run_seed_fn.py wraps the replay in yarr's PyTorchReplayBuffer (:131): `.dataset()` gives the iterator the
    runner pulls batches from, `.replay_buffer` the store it shuts down at the end (offline_train_runner.py:130, :172)."""

    def __init__(self, replay_buffer, num_workers=0, device=None):
        self.replay_buffer, self._device = replay_buffer, device

    def dataset(self, batch_size=None, drop_last=False):
        import torch
        if self._device is not None and torch.cuda.is_available():
            return DeviceBatchStream(self.replay_buffer, self._device, batch_size=batch_size)
        return _HostBatches(self.replay_buffer, batch_size)


class _HostBatches:
    def __init__(self, replay, batch_size):
        self._replay, self._b = replay, batch_size

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        b = self._replay.sample_transition_batch(self._b)
        return {k: torch.from_numpy(v) for k, v in b.items() if v.dtype != object}
