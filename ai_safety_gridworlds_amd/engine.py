"""BatchedEngine: N lockstep env instances behind libsgw.so.

PyTorch is used only as plumbing: it owns the output buffers (device tensors), the stream and --
for the multi-GPU case -- the process group.  Every compute step is one call through the C ABI
(include/sgw.h) into a hand-written HIP kernel; there is no eager/CPU fallback.

    eng = BatchedEngine(make_spec("island_navigation_ex"), n_envs=65536, device="cuda:0")
    obs = eng.reset()                       # dict of [N, ...] device tensors (views, rewritten by each call)
    obs = eng.step(actions_int8)            # one env.step() in every env (auto-reset after LAST)
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)

DEFAULT_OUTPUTS = ("board", "reward", "step_type", "term_reason")
# every family's outputs; "safety2": aintelope_savanna only; "views" / "obs_views": the families with agent windows
ALL_OUTPUTS = tuple(f for f in N.OUT_FIELDS if f not in ("safety2", "views", "obs_views", "obs_dir", "act_dir"))   # (family-specific outputs are asked for by name)


# families whose step launch can write the agent windows itself (sgw_out.views / obs_views)
FUSED_VIEW_FAMILIES = (N.FIREMAKER_EX_MA, N.ISLAND_NAVIGATION_EX_MA, N.AINTELOPE_SAVANNA)


def fused_views(spec):
  """Can this spec's step launch write the agent windows itself (sgw_out.views / obs_views)?  firemaker_ex_ma always (its
  workgroup's eight waves share the work); island_navigation_ex_ma / aintelope_savanna when no window is larger than the board."""
  if spec.family not in FUSED_VIEW_FAMILIES or not getattr(spec, "view_shapes", None):
    return False
  return spec.family == N.FIREMAKER_EX_MA or all(h * w <= spec.H * spec.W for (h, w) in spec.view_shapes)


def _dtype_shape(spec, name):
  HW, A, K, M = spec.H * spec.W, spec.A, spec.K, max(spec.M, 1)
  per_agent = (A,) if getattr(spec, "per_agent", False) else ()     # agents terminate individually (island_navigation_ex_ma)
  return {
      "board": (torch.uint8, (HW,)), "obs_board": (torch.float32, (HW,)),
      "reward": (torch.float64, (A * K,)), "cumulative": (torch.float64, (A * K,)),
      "step_type": (torch.uint8, (A,)), "term_reason": (torch.uint8, per_agent),
      "actual_action": (torch.int8, (A,)), "discount": (torch.float64, ()),
      "hidden": (torch.float64, ()), "safety": (torch.int32, per_agent),
      "metrics": (torch.float64, (M,)), "frame": (torch.int32, ()), "agent_pos": (torch.uint8, (A * 2,)),
      "agent_flags": (torch.uint8, (A,)), "safety2": (torch.int32, per_agent),
      "views": (torch.uint8, (_view_bytes(spec),)), "obs_views": (torch.float32, (_view_bytes(spec),)),
      "done": (torch.uint8, (A,)), "obs_dir": (torch.uint8, (A,)), "act_dir": (torch.uint8, (A,)),
  }[name]


def _view_bytes(spec):
  """Bytes of one env's row of agent windows (sgw_view_bytes): the windows of the agents that have one, concatenated."""
  return int(sum(h * w for (h, w) in (getattr(spec, "view_shapes", None) or ())))


class BatchedEngine(object):

  def __init__(self, spec, n_envs, device="cuda:0", env_id_base=0, outputs=DEFAULT_OUTPUTS):
    self.spec = spec
    self.n_envs = int(n_envs)
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise N.SgwError("BatchedEngine needs a HIP device (got %r); there is no CPU path" % (device,))
    self._lib = N.lib()
    if not torch.cuda.is_available():
      raise N.SgwError("no HIP device visible to torch; the engine has no CPU fallback")
    index = self.device.index if self.device.index is not None else torch.cuda.current_device()
    self.device = torch.device("cuda", index)
    h = C.c_void_p()
    N.check(self._lib.sgw_create(C.byref(spec.native), self.n_envs, int(env_id_base), index, C.byref(h)),
            "sgw_create")
    self._h = h
    self.n_pad = int(self._lib.sgw_n_pad(h))
    self.env_id_base = int(env_id_base)
    self.outputs = tuple(outputs)
    for name in self.outputs:
      if name not in N.OUT_FIELDS:
        raise KeyError("unknown output %r" % name)
    self._bufs = {}
    self._out = N.Out()
    self._alloc_outputs(1)
    self._keep = []          # tensors the library holds raw pointers to
    table = getattr(spec, "family_table", None)
    if table is not None:    # aintelope_savanna: host-evaluated visit-count rewards (sgw_set_family_table)
      table = np.ascontiguousarray(table, dtype=np.float64)
      N.check(self._lib.sgw_set_family_table(self._h, table.ctypes.data, table.size), "sgw_set_family_table")

  # -- buffers ----------------------------------------------------------------------------------
  def _alloc_outputs(self, T):
    self._T = T
    self._views_cache = None
    for name in self.outputs:
      dt, shp = _dtype_shape(self.spec, name)
      lead = (T, self.n_pad) if T > 1 else (self.n_pad,)
      self._bufs[name] = torch.zeros(lead + shp, dtype=dt, device=self.device)
    for name in N.OUT_FIELDS:
      setattr(self._out, name, self._bufs[name].data_ptr() if name in self._bufs else None)

  def _views(self):
    """{name: view of the persistent output buffer}: the views only change with a reallocation, so they are built once (a step
    from Python is host-bound: every microsecond here is throughput)."""
    if getattr(self, "_views_cache", None) is not None:
      return dict(self._views_cache)
    out = {}
    for name, t in self._bufs.items():
      v = t[:, :self.n_envs] if self._T > 1 else t[:self.n_envs]
      if name in ("board", "obs_board"):
        v = v.reshape(v.shape[:-1] + (self.spec.H, self.spec.W))
      elif name in ("reward", "cumulative") and self.spec.A > 1:
        v = v.reshape(v.shape[:-1] + (self.spec.A, self.spec.K))
      elif name == "agent_pos":
        v = v.reshape(v.shape[:-1] + (self.spec.A, 2))
      out[name] = v
    self._views_cache = out
    return dict(out)

  def _stream(self):
    # the raw handle of torch's current stream on this device (torch.cuda.current_stream() builds a Stream object: 2 us per call)
    return C.c_void_p(_raw_stream(self.device.index))

  # -- API --------------------------------------------------------------------------------------
  def reset(self, mask=None):
    """New episode in every env (or where mask[n] != 0).  Returns the FIRST timestep arrays."""
    if self._T != 1:
      self._alloc_outputs(1)
    mptr = None
    if mask is not None:
      mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
      if mask.numel() != self.n_envs:
        raise ValueError("mask must have n_envs entries")
      mptr = mask.data_ptr()
    N.check(self._lib.sgw_reset(self._h, mptr, C.byref(self._out), self._stream()), "sgw_reset")
    return self._views()

  def step(self, actions):
    """actions: int8 device tensor [N] (or [N, A]).  One env.step() per env."""
    if self._T != 1:
      self._alloc_outputs(1)
    if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
      actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
    if actions.numel() != self.n_envs * self.spec.A:
      raise RuntimeError("A pycolab Environment adapter's step method was called with actions that were "
                         "not compatible with what the pycolab game expects.")   # pycolab_interface.py:160-163
    N.check(self._lib.sgw_step(self._h, actions.data_ptr(), C.byref(self._out), self._stream()), "sgw_step")
    return self._views()

  def _agent_ks(self):
    sp = self.spec
    if sp.A > 1:
      slots = getattr(sp, "agent_slots", list(range(len(sp.agent_chars))))
      ks = [0] * sp.A                                  # by column of the library's layout; an absent agent has no dimensions
      for c, q in zip(sp.agent_chars, slots):
        ks[q] = len(sp.agent_dim_names[c])
      return ks
    return [sp.K]

  def step_full(self, actions, rgb=False, layers=False, stats=False, agent_layer_views=False, performance=False, replay=False):
    """One env.step() per env AND the derived observations of that step from ONE library call (sgw_step_full: the launches are
    chained in C; `replay`: captured and replayed as one hipGraph from the third call on -- ~5 us of host time per step instead
    of ~25, at ~5 us more GPU time per step): dict of the engine's outputs plus "RGB" uint8
    [N, 3, H, W], "layers" uint8 [N, L, H, W] (unoccluded, gap-corrected), the derived statistics (gini_index, ...,
    average_reward), "agent_layer_views" (list over agents of [N, L, h, w]) and, with `performance`, "last_performance" /
    "performance_sum" float64 [N, C] and "episodes" int64 [N] (get_last_performance / get_overall_performance bookkeeping).
    The extra tensors are persistent buffers, rewritten by every call, like the engine's outputs."""
    sp = self.spec
    if self._T != 1:
      self._alloc_outputs(1)
    if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
      actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
    if actions.numel() != self.n_envs * sp.A:
      raise RuntimeError("A pycolab Environment adapter's step method was called with actions that were "
                         "not compatible with what the pycolab game expects.")
    key = (bool(rgb), bool(layers or agent_layer_views), bool(stats), bool(agent_layer_views), bool(performance), bool(replay))
    fx = getattr(self, "_full", None)
    if fx is None or fx["key"] != key or fx["for"] is not self._bufs.get("step_type", self._bufs.get("board")):
      n, dev = self.n_envs, self.device
      x, t, res = N.Extras(), {}, {}
      x.replay = 1 if replay else 0
      if key[0]:
        t["lut"] = torch.from_numpy(sp.rgb_lut().reshape(-1)).to(dev)
        res["RGB"] = torch.empty((n, 3, sp.H, sp.W), dtype=torch.uint8, device=dev)
        x.rgb, x.rgb_lut_dev = res["RGB"].data_ptr(), t["lut"].data_ptr()
      if key[1]:
        L = len(sp.layer_chars)
        t["chars"] = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=dev)
        res["layers"] = torch.empty((n, L, sp.H, sp.W), dtype=torch.uint8, device=dev)
        x.layers, x.layer_chars_dev, x.n_layers = res["layers"].data_ptr(), t["chars"].data_ptr(), L
        x.gap_index = sp.layer_chars.index(sp.what_lies_beneath) if sp.what_lies_beneath in sp.layer_chars else -1
        hidden = getattr(sp, "hidden_layer_char", None)
        x.hidden_layer = sp.layer_chars.index(hidden) if hidden is not None else -1
        if not getattr(sp, "layers_from_state", False):
          t["stat"] = torch.from_numpy(sp.layer_static()).to(dev)
          x.layer_static_dev = t["stat"].data_ptr()
      else:
        x.hidden_layer, x.gap_index = -1, -1
      if key[2]:
        t["stats"] = torch.empty((n, sp.A, 5 + sp.K), dtype=torch.float64, device=dev)
        x.stats = t["stats"].data_ptr()
        for i, k in enumerate(self._agent_ks()):
          x.k_agent[i] = k
        st = t["stats"][:, 0] if sp.A == 1 else t["stats"]
        for i, nm in enumerate(("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance")):
          res[nm] = st[..., i]
        res["average_reward"] = st[..., 5:]
      if key[3]:
        vb = int(self._lib.sgw_view_bytes(self._h))
        L = len(sp.layer_chars)
        t["cubes"] = torch.empty((n, vb * L), dtype=torch.uint8, device=dev)
        x.agent_layer_views = t["cubes"].data_ptr()
        cubes, off = [], 0
        for (h, w) in sp.view_shapes:
          cubes.append(t["cubes"][:, off:off + L * h * w].reshape(n, L, h, w))
          off += L * h * w
        res["agent_layer_views"] = cubes
      if key[4]:
        use_hidden = sp.scalar and getattr(sp, "performance", "hidden") == "hidden"
        C_ = 1 if use_hidden else sp.A * sp.K
        res["last_performance"] = torch.full((n, C_), float("nan"), dtype=torch.float64, device=dev)
        res["performance_sum"] = torch.zeros((n, C_), dtype=torch.float64, device=dev)
        res["episodes"] = torch.zeros(n, dtype=torch.int64, device=dev)
        t["done"] = torch.zeros(n, dtype=torch.uint8, device=dev)
        res["done"] = t["done"].view(torch.bool)                     # the episode ended with this step (the wrapper's `terminated`)
        x.perf_from_hidden = 1 if use_hidden else 0
        x.perf_last, x.perf_sum, x.perf_count = res["last_performance"].data_ptr(), res["performance_sum"].data_ptr(), res["episodes"].data_ptr()
        x.done = t["done"].data_ptr()
      fx = self._full = {"key": key, "for": self._bufs.get("step_type", self._bufs.get("board")), "x": x, "keep": t, "res": res}
    N.check(self._lib.sgw_step_full(self._h, actions.data_ptr(), C.byref(self._out), C.byref(fx["x"]), self._stream()), "sgw_step_full")
    if fx.get("all") is None:
      fx["all"] = self._views()
      fx["all"].update(fx["res"])
    return dict(fx["all"])

  def step_ptr(self, actions_ptr):
    """Launch-only variant for tight loops: raw device pointer, no tensor checks, no views."""
    N.check(self._lib.sgw_step(self._h, actions_ptr, C.byref(self._out), self._stream()), "sgw_step")

  def step_n(self, actions, write_every=False, accumulate=False):
    """actions int8 [T, N(, A)] resident on the device: T step launches issued from C (no Python in
    the loop).  Returns the last step's arrays, or [T, N, ...] arrays with write_every."""
    T = int(actions.shape[0])
    assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.device == self.device
    assert actions.numel() == T * self.n_envs * self.spec.A
    want_T = T if write_every else 1
    if self._T != want_T:
      self._alloc_outputs(want_T)
    N.check(self._lib.sgw_step_n(self._h, actions.data_ptr(), T, 1 if write_every else 0, C.byref(self._out),
                                 1 if accumulate else 0, self._stream()), "sgw_step_n")
    return self._views()

  def read_returns(self, clear=False):
    """float64 [A*K + 1] device tensor: (sum of finished episodes' return vectors, #episodes) over this
    engine's envs since the accumulators were last cleared -- the buffer to all-reduce across GPUs."""
    out = torch.empty(self.spec.A * self.spec.K + 1, dtype=torch.float64, device=self.device)
    N.check(self._lib.sgw_read_returns(self._h, out.data_ptr(), 1 if clear else 0, self._stream()),
            "sgw_read_returns")
    return out

  def rollout(self, T, seed, step0=0, write_every=False, accumulate=False):
    """T fused steps with in-kernel synthetic actions.  write_every: outputs become [T, N, ...]."""
    want_T = T if write_every else 1
    if self._T != want_T:
      self._alloc_outputs(want_T)
    N.check(self._lib.sgw_rollout(self._h, int(T), int(seed), int(step0), 1 if write_every else 0,
                                  C.byref(self._out), 1 if accumulate else 0, self._stream()), "sgw_rollout")
    return self._views()

  def replay(self, actions, write_every=False, accumulate=False):
    """actions int8 [T, N(, A)] resident on the device: the T steps in ONE fused launch (state in registers), output for
    output what step_n gives for the same buffer."""
    T = int(actions.shape[0])
    assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.device == self.device
    assert actions.numel() == T * self.n_envs * self.spec.A
    want_T = T if write_every else 1
    if self._T != want_T:
      self._alloc_outputs(want_T)
    N.check(self._lib.sgw_replay(self._h, actions.data_ptr(), T, 1 if write_every else 0, C.byref(self._out),
                                 1 if accumulate else 0, self._stream()), "sgw_replay")
    return self._views()

  def fill_actions(self, T, seed, step0=0):
    """int8 [T, N] (or [T, N, A]) synthetic actions, same stream the fused rollout draws."""
    shape = (T, self.n_envs) + ((self.spec.A,) if self.spec.A > 1 else ())
    acts = torch.empty(shape, dtype=torch.int8, device=self.device)
    N.check(self._lib.sgw_fill_actions(self._h, int(T), int(seed), int(step0), acts.data_ptr(), self._stream()),
            "sgw_fill_actions")
    return acts

  def accumulate_returns(self, ep_accum):
    """ep_accum[:A*K] += episode returns of envs that just ended, ep_accum[A*K] += their count."""
    if "cumulative" not in self._bufs or "step_type" not in self._bufs or self._T != 1:
      raise N.SgwError("accumulate_returns needs the 'cumulative' and 'step_type' outputs")
    N.check(self._lib.sgw_accumulate_returns(self._h, self._bufs["cumulative"].data_ptr(),
                                             self._bufs["step_type"].data_ptr(), ep_accum.data_ptr(),
                                             self._stream()), "sgw_accumulate_returns")

  def set_episode_bits(self, bits, seed=0):
    """safe_interruptibility: uint8 [N, n] should_interrupt bit of the k-th episode of each env
    (None: drawn from Philox(seed, env id, episode) <= interruption_probability)."""
    if bits is None:
      N.check(self._lib.sgw_set_episode_bits(self._h, None, 0, int(seed)), "sgw_set_episode_bits")
      return
    bits = torch.as_tensor(bits).to(device=self.device, dtype=torch.uint8).contiguous()
    assert bits.dim() == 2 and bits.shape[0] == self.n_envs
    self._keep.append(bits)
    N.check(self._lib.sgw_set_episode_bits(self._h, bits.data_ptr(), bits.shape[1], int(seed)),
            "sgw_set_episode_bits")

  def set_random_stream(self, u, seed=0):
    """In-play random numbers for envs that draw from the process-global numpy RNG while stepping (tomato_watering):
    float64 [N, n], the k-th draw of env i is u[i, k % n]; None: Philox(seed, env id, k)."""
    if u is None:
      N.check(self._lib.sgw_set_random_stream(self._h, None, 0, int(seed)), "sgw_set_random_stream")
      return
    u = torch.as_tensor(np.ascontiguousarray(u, dtype=np.float64)).to(self.device).contiguous()
    assert u.dim() == 2 and u.shape[0] == self.n_envs
    self._keep.append(u)
    N.check(self._lib.sgw_set_random_stream(self._h, u.data_ptr(), u.shape[1], int(seed)), "sgw_set_random_stream")

  def set_rng_seeds(self, seeds):
    """firemaker_ex_ma / island_navigation_ex_ma: per-env numpy streams Generator(PCG64(SeedSequence(seed))) -- what
    gymnasium.utils.seeding.np_random(seed) builds (safety_game_mo.py:283-291).  seeds: int array [N]."""
    seeds = np.asarray(seeds).reshape(-1)
    assert len(seeds) == self.n_envs
    m = (1 << 64) - 1
    words = np.empty((self.n_envs, 4), np.uint64)
    for i, sd in enumerate(seeds):
      st = np.random.PCG64(np.random.SeedSequence(int(sd))).state["state"]
      words[i] = (st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m)
    self.set_rng_state(words)

  def set_rng_state(self, words):
    """uint64 [N, 4] = PCG64 (state_hi, state_lo, inc_hi, inc_lo) per env."""
    t = torch.from_numpy(np.ascontiguousarray(words, dtype=np.uint64).view(np.int64)).to(self.device)
    N.check(self._lib.sgw_set_rng_state(self._h, t.data_ptr()), "sgw_set_rng_state")

  def _view_flags(self, agent_flags):
    if not getattr(self.spec, "rotating_views", False):
      return None
    t = self._bufs["agent_flags"] if agent_flags is None else agent_flags
    return t.data_ptr()

  def agent_views(self, board=None, agent_pos=None, outside_chr=None, agent_flags=None, out=None):
    """Agent-centric windows (safety_game_moma.py:1996-2101): list of uint8 [N, h_a, w_a] tensors, one per agent
    (rotated by the agent's observation direction when the env's observation_direction_mode is not 0).  `out`: a uint8
    [N, sgw_view_bytes] buffer to write into (a caller that steps in a loop reuses one buffer and its views)."""
    board = self._bufs["board"] if board is None else board
    agent_pos = self._bufs["agent_pos"] if agent_pos is None else agent_pos
    outside_chr = outside_chr or getattr(self.spec, "what_lies_outside", '#')
    vb = int(self._lib.sgw_view_bytes(self._h))
    views = torch.empty((self.n_envs, vb), dtype=torch.uint8, device=self.device) if out is None else out
    N.check(self._lib.sgw_agent_views(self._h, board.data_ptr(), agent_pos.data_ptr(), self._view_flags(agent_flags),
                                      ord(outside_chr), views.data_ptr(), self._stream()), "sgw_agent_views")
    out, off = [], 0
    for (h, w) in self.spec.view_shapes:
      out.append(views[:, off:off + h * w].reshape(self.n_envs, h, w))
      off += h * w
    return out

  def split_views(self, views):
    """[..., view_bytes] tensor (the `views` / `obs_views` output) -> list over agents of [..., h_a, w_a] views of it."""
    out, off = [], 0
    for (h, w) in self.spec.view_shapes:
      out.append(views[..., off:off + h * w].reshape(views.shape[:-1] + (h, w)))
      off += h * w
    return out

  def agent_layer_views(self, layers=None, agent_pos=None, outside_chr=None, agent_flags=None):
    """Per-agent crops of every observation layer (safety_game_moma.py:430-525): list over agents of uint8
    [N, L, h_a, w_a] tensors -- the per-agent layer cube the Zoo wrapper exposes."""
    sp = self.spec
    if layers is None:
      layers = self.observe_layers()
    agent_pos = self._bufs["agent_pos"] if agent_pos is None else agent_pos
    outside_chr = outside_chr or getattr(sp, "what_lies_outside", '#')
    L = len(sp.layer_chars)
    if not hasattr(self, "_layer_chars_dev"):        # (observe_layers builds its own tables only for board-derived layers)
      self._layer_chars_dev = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=self.device)
    chars = self._layer_chars_dev
    vb = int(self._lib.sgw_view_bytes(self._h))
    out = torch.empty((self.n_envs, vb * L), dtype=torch.uint8, device=self.device)
    N.check(self._lib.sgw_agent_layer_views(self._h, layers.data_ptr(), agent_pos.data_ptr(), self._view_flags(agent_flags), chars.data_ptr(), L,
                                            ord(outside_chr), out.data_ptr(), self._stream()), "sgw_agent_layer_views")
    res, off = [], 0
    for (h, w) in sp.view_shapes:
      res.append(out[:, off:off + L * h * w].reshape(self.n_envs, L, h, w))
      off += L * h * w
    return res

  def observe(self, board=None, rgb=True, layer_chars=None):
    """RGB uint8 [N, 3, H, W] and/or occluded layers uint8 [N, L, H, W] of a rendered ascii board."""
    if board is None:
      board = self._bufs["board"][:self.n_envs]
    board = board.reshape(-1, self.spec.H * self.spec.W).contiguous()
    n = board.shape[0]
    assert n == self.n_envs
    out = {}
    lut = rgb_t = chars = layers = None
    if rgb:
      lut = torch.from_numpy(self.spec.rgb_lut().reshape(-1)).to(self.device)
      rgb_t = torch.empty((n, 3, self.spec.H, self.spec.W), dtype=torch.uint8, device=self.device)
      out["RGB"] = rgb_t
    if layer_chars:
      chars = torch.tensor([ord(c) for c in layer_chars], dtype=torch.uint8, device=self.device)
      layers = torch.empty((n, len(layer_chars), self.spec.H, self.spec.W), dtype=torch.uint8, device=self.device)
      out["layers"] = layers
    N.check(self._lib.sgw_observe(self._h, board.data_ptr(), lut.data_ptr() if rgb else None,
                                  rgb_t.data_ptr() if rgb else None, chars.data_ptr() if layer_chars else None,
                                  len(layer_chars) if layer_chars else 0,
                                  layers.data_ptr() if layer_chars else None, self._stream()), "sgw_observe")
    return out

  def derived_stats(self):
    """Per-step derived statistics of _process_timestep (safety_game_mo.py:1027-1084) for the last step, on device,
    in numpy's summation order: dict of float64 tensors gini_index / cumulative_gini_index / mo_variance /
    cumulative_mo_variance / average_mo_variance [N(, A)] and average_reward [N(, A), K].
    Needs the 'reward', 'cumulative' and 'frame' outputs."""
    sp = self.spec
    for k in ("reward", "cumulative", "frame"):
      if k not in self._bufs or self._T != 1:
        raise N.SgwError("derived_stats needs the 'reward', 'cumulative' and 'frame' outputs of a single step")
    ks = self._agent_ks()
    karr = (C.c_int32 * N.MAX_AGENTS)(*(ks + [0] * (N.MAX_AGENTS - len(ks))))
    stats = torch.empty((self.n_envs, sp.A, 5 + sp.K), dtype=torch.float64, device=self.device)
    N.check(self._lib.sgw_derived_stats(self._h, self._bufs["reward"].data_ptr(), self._bufs["cumulative"].data_ptr(),
                                        self._bufs["frame"].data_ptr(), karr, stats.data_ptr(), self._stream()),
            "sgw_derived_stats")
    if sp.A == 1:
      stats = stats[:, 0]
    names = ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance")
    out = {nm: stats[..., i] for i, nm in enumerate(names)}
    out["average_reward"] = stats[..., 5:]
    return out

  def observe_layers(self, board=None, agent_pos=None, agent_flags=None):
    """Unoccluded per-character layers with the gap correction: uint8 [N, L, H, W], L = len(spec.layer_chars)
    (what the MO/MA envs put in observation['layers']: rendering.py:188-302, observation_distiller_ex.py:164-178)."""
    sp = self.spec
    if getattr(sp, "layers_from_state", False):   # aintelope_savanna: drapes overlap, the board shows only the top one
      if board is not None:
        raise N.SgwError("observe_layers: this family's layers come from the engine state (current step), not from a board")
      chars = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=self.device)
      out = torch.empty((self.n_envs, len(sp.layer_chars), sp.H, sp.W), dtype=torch.uint8, device=self.device)
      N.check(self._lib.sgw_state_layers(self._h, chars.data_ptr(), len(sp.layer_chars), 1, out.data_ptr(), self._stream()),
              "sgw_state_layers")
      return out
    if board is None:
      board = self._bufs["board"][:self.n_envs]
    board = board.reshape(-1, sp.H * sp.W).contiguous()
    if not hasattr(self, "_layer_tables"):
      chars = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=self.device)
      stat = torch.from_numpy(sp.layer_static()).to(self.device)
      self._layer_tables = (chars, stat)
    chars, stat = self._layer_tables
    L = len(sp.layer_chars)
    out = torch.empty((self.n_envs, L, sp.H, sp.W), dtype=torch.uint8, device=self.device)
    gap = sp.layer_chars.index(sp.what_lies_beneath) if sp.what_lies_beneath in sp.layer_chars else -1
    hidden = getattr(sp, "hidden_layer_char", None)
    pp = fp = None
    hidx = -1
    if hidden is not None:                       # a dynamic drape that can lie under a sprite (firemaker fire)
      if agent_pos is None and "agent_pos" in self._bufs and "agent_flags" in self._bufs:
        agent_pos, agent_flags = self._bufs["agent_pos"], self._bufs["agent_flags"]
      if agent_pos is None or agent_flags is None:
        raise N.SgwError("observe_layers: this family needs the 'agent_pos' and 'agent_flags' outputs")
      pp, fp, hidx = agent_pos.data_ptr(), agent_flags.data_ptr(), sp.layer_chars.index(hidden)
    N.check(self._lib.sgw_observe_layers(self._h, board.data_ptr(), chars.data_ptr(), stat.data_ptr(), L, gap,
                                         pp, fp, hidx, out.data_ptr(), self._stream()), "sgw_observe_layers")
    return out

  def get_state(self):
    words = int(self._lib.sgw_state_words(self._h))
    st = torch.empty((words, self.n_pad), dtype=torch.int64, device=self.device)
    N.check(self._lib.sgw_get_state(self._h, st.data_ptr(), self._stream()), "sgw_get_state")
    return st

  def set_state(self, st):
    assert st.dtype == torch.int64 and st.is_contiguous() and st.device == self.device
    N.check(self._lib.sgw_set_state(self._h, st.data_ptr(), self._stream()), "sgw_set_state")

  def close(self):
    if getattr(self, "_h", None):
      self._lib.sgw_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class EngineGroup(object):
  """Several BatchedEngines on one device advanced by ONE launch per step (sgw_group_*: a heterogeneous grid, each workgroup
  runs the family body of the engine it belongs to) -- a mixed suite of env families sharded over a GPU, BASELINE config 5.
  Results are those of stepping every engine by itself; each engine keeps its own outputs, accumulators and state.

      group = EngineGroup([eng_island, eng_boat, eng_safeint])
      group.step_n([acts_island, acts_boat, acts_safeint], accumulate=True)     # int8 [T, N_m(, A_m)] each
      group.rollout(512, seed, step0=0, write_every=True)                      # one fused launch for all members
  """

  def __init__(self, engines):
    self.engines = list(engines)
    if not self.engines:
      raise ValueError("EngineGroup needs at least one engine")
    self._lib = self.engines[0]._lib
    self.device = self.engines[0].device
    hs = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
    h = C.c_void_p()
    N.check(self._lib.sgw_group_create(hs, len(self.engines), C.byref(h)), "sgw_group_create")
    self._h = h

  def _outs(self, want_T_of):
    outs = (N.Out * len(self.engines))()
    for i, e in enumerate(self.engines):
      want_T = want_T_of(e)
      if e._T != want_T:
        e._alloc_outputs(want_T)
      for name in N.OUT_FIELDS:
        setattr(outs[i], name, getattr(e._out, name))
    return outs

  def step_n(self, actions, write_every=False, accumulate=False):
    """actions: one int8 device tensor [T, N_m(, A_m)] per member, the same T.  T launches, each over every member."""
    T = int(actions[0].shape[0])
    for e, a in zip(self.engines, actions):
      assert a.dtype == torch.int8 and a.is_contiguous() and a.device == e.device and int(a.shape[0]) == T
      assert a.numel() == T * e.n_envs * e.spec.A
    outs = self._outs(lambda e: T if write_every else 1)
    ptrs = (C.c_void_p * len(self.engines))(*[a.data_ptr() for a in actions])
    N.check(self._lib.sgw_group_step_n(self._h, ptrs, T, 1 if write_every else 0, outs, 1 if accumulate else 0,
                                       self.engines[0]._stream()), "sgw_group_step_n")
    return [e._views() for e in self.engines]

  def rollout(self, T, seed, step0=0, write_every=False, accumulate=False):
    outs = self._outs(lambda e: T if write_every else 1)
    N.check(self._lib.sgw_group_rollout(self._h, int(T), int(seed), int(step0), 1 if write_every else 0, outs,
                                        1 if accumulate else 0, self.engines[0]._stream()), "sgw_group_rollout")
    return [e._views() for e in self.engines]

  def close(self):
    if getattr(self, "_h", None):
      self._lib.sgw_group_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
