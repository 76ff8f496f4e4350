"""CSV / gzip step logger with the reference's file format (shared/safety_game_mo.py:727-820 header,
1110-1227 rows): `;`-separated, one row per logged env and step with the_plot.frame > 0, numbers written through a
10-digit ROUND_HALF_UP decimal context with exponents removed.  Host-side observability: the numbers come from the
engine's outputs and its device-side derived statistics (sgw_derived_stats, bit-exact with numpy's reduction order).

    env = BatchedSafetyEnvironment("island_navigation_ex", num_envs=4096)
    log = StepLogger(env, [LOG_EPISODE, LOG_ITERATION, LOG_REWARD, LOG_GINI_INDEX, LOG_METRICS], log_dir="logs", env_indices=[0, 17])
    ts = env.reset(); log.on_reset(); log.write(ts)
    ts = env.step(actions); log.write(ts)
"""
import csv
import datetime
import decimal
import gzip
import numbers
import os

import numpy as np

LOG_TIMESTAMP = 'timestamp'                    # safety_game_mo.py:84-105
LOG_ENVIRONMENT = 'env'
LOG_TRIAL = 'trial'
LOG_ENV_LAYOUT_SEED = 'env layout seed'
LOG_ENV_SEED = 'env seed'
LOG_EPISODE = 'episode'
LOG_ITERATION = 'iteration'
LOG_ARGUMENTS = 'arguments'
LOG_REWARD = 'reward'
LOG_SCALAR_REWARD = 'scalar_reward'
LOG_CUMULATIVE_REWARD = 'cumulative_reward'
LOG_AVERAGE_REWARD = 'average_reward'
LOG_GINI_INDEX = 'gini_index'
LOG_CUMULATIVE_GINI_INDEX = 'cumulative_gini_index'
LOG_MO_VARIANCE = 'mo_variance'
LOG_CUMULATIVE_MO_VARIANCE = 'cumulative_mo_variance'
LOG_AVERAGE_MO_VARIANCE = 'average_mo_variance'
LOG_SCALAR_CUMULATIVE_REWARD = 'scalar_cumulative_reward'
LOG_SCALAR_AVERAGE_REWARD = 'scalar_average_reward'
LOG_METRICS = 'metric'

LOG_COMPRESSLEVEL = 6                          # safety_game_mo.py:58
_CTX = decimal.Context(prec=10, rounding=decimal.ROUND_HALF_UP, capitals=0)    # safety_game_mo.py:398-400


def format_float(value):
  """safety_game_mo.py:1218-1227."""
  if isinstance(value, numbers.Number):
    d = _CTX.create_decimal_from_float(float(value))
    integral = d.to_integral()
    return integral if d == integral else d.normalize()
  return str(value)


class StepLogger(object):

  def __init__(self, env, log_columns, log_dir="logs", log_filename=None, log_filename_comment="", gzip_log=False,
               env_indices=(0,), trial=1, env_seed=None, log_arguments=None):
    self.env, self.spec = env, env.spec
    if self.spec.A != 1 or self.spec.scalar:
      raise NotImplementedError("StepLogger covers the single-agent multi-objective envs (SafetyEnvironmentMo's logger)")
    for need in ("reward", "cumulative", "frame", "metrics"):
      if need not in env.engine.outputs:
        raise ValueError("StepLogger needs the %r output" % need)
    self.columns = list(log_columns)
    self.env_indices = [int(i) for i in env_indices]
    self.trial, self.env_seed = trial, env_seed
    self.log_arguments = {} if log_arguments is None else log_arguments
    self.metric_order = list(getattr(self.spec, "metrics_log_order", self.spec.metric_names))
    self._metric_cols = [self.spec.metric_names.index(m) for m in self.metric_order]
    self.episode = {i: 1 for i in self.env_indices}
    self._played = {i: False for i in self.env_indices}
    if log_dir and not os.path.exists(log_dir):
      os.makedirs(log_dir)
    if log_filename is None:
      stamp = datetime.datetime.strftime(datetime.datetime.now(), '%Y.%m.%d-%H.%M.%S')
      log_filename = ("ai_safety_gridworlds_amd." + self.spec.name + ("-" if log_filename_comment else "") + log_filename_comment
                      + "-" + stamp + ".csv")                                             # safety_game_mo.py:596-601
    self.path = os.path.join(log_dir, log_filename + (".gz" if gzip_log else ""))
    if gzip_log:
      self._file = gzip.open(self.path, mode='wt', newline='', encoding='utf-8', compresslevel=LOG_COMPRESSLEVEL)
    else:
      self._file = open(self.path, mode='wt', buffering=1024 * 1024, newline='', encoding='utf-8')
    self._writer = csv.writer(self._file, quoting=csv.QUOTE_MINIMAL, delimiter=';')
    self._writer.writerow(self._header())

  def _header(self):
    dims, data = self.spec.dim_names, []
    for col in self.columns:
      if col in (LOG_REWARD, LOG_CUMULATIVE_REWARD, LOG_AVERAGE_REWARD):
        data += [col + "_" + d for d in dims]
      elif col == LOG_METRICS:
        data += [LOG_METRICS + "_" + m for m in self.metric_order]
      else:
        data.append(col)
    return data

  def on_reset(self, mask=None):
    """Call after an EXPLICIT env.reset(): the episode counter advances for envs whose episode had a step
    (safety_game_mo.py:690-700); auto-resets inside step() do not advance it."""
    for i in self.env_indices:
      if (mask is None or bool(mask[i])) and self._played[i]:
        self.episode[i] += 1
      if mask is None or bool(mask[i]):
        self._played[i] = False

  def write(self, timestep):
    o = timestep.observation
    idx = self.env_indices
    frame = o["frame"][idx].cpu().numpy()
    K, M = self.spec.K, self.spec.M
    reward = o["reward"].reshape(-1, K)[idx].cpu().numpy()
    cumulative = o["cumulative"].reshape(-1, K)[idx].cpu().numpy()
    metrics = o["metrics"][idx][:, :M].cpu().numpy()
    stats = {k: v[idx].cpu().numpy() for k, v in self.env.engine.derived_stats().items()}
    for j, i in enumerate(idx):
      it = int(frame[j])
      if it > 0:
        self._played[i] = True
      else:
        continue                                   # safety_game_mo.py:1087: rows only when the_plot.frame > 0
      r = [float(x) for x in reward[j]]
      c = [float(x) for x in cumulative[j]]
      avg = [x / (it + 1) for x in c]              # safety_game_mo.py:1030
      data = []
      for col in self.columns:
        if col == LOG_TIMESTAMP:
          data.append(datetime.datetime.strftime(datetime.datetime.now(), '%Y.%m.%d-%H.%M.%S'))
        elif col == LOG_ENVIRONMENT:
          data.append("ai_safety_gridworlds_amd." + self.spec.name)
        elif col == LOG_ENV_SEED:
          data.append(self.env_seed)
        elif col in (LOG_ENV_LAYOUT_SEED, LOG_TRIAL):
          data.append(self.trial)
        elif col == LOG_EPISODE:
          data.append(self.episode[i])
        elif col == LOG_ITERATION:
          data.append(it)
        elif col == LOG_ARGUMENTS:
          data.append(str(self.log_arguments))
        elif col == LOG_REWARD:
          data += [format_float(x) for x in r]
        elif col == LOG_SCALAR_REWARD:
          data.append(format_float(sum(r)))
        elif col == LOG_CUMULATIVE_REWARD:
          data += [format_float(x) for x in c]
        elif col == LOG_AVERAGE_REWARD:
          data += [format_float(x) for x in avg]
        elif col == LOG_SCALAR_CUMULATIVE_REWARD:
          data.append(format_float(sum(c)))
        elif col == LOG_SCALAR_AVERAGE_REWARD:
          data.append(format_float(sum(avg)))
        elif col in (LOG_GINI_INDEX, LOG_CUMULATIVE_GINI_INDEX, LOG_MO_VARIANCE, LOG_CUMULATIVE_MO_VARIANCE, LOG_AVERAGE_MO_VARIANCE):
          data.append(format_float(float(stats[col][j])))
        elif col == LOG_METRICS:
          data += [format_float(float(metrics[j][k])) for k in self._metric_cols]
        else:
          raise KeyError("unknown log column %r" % col)
      self._writer.writerow(data)

  def flush(self):
    self._file.flush()

  def close(self):
    if self._file:
      self._file.flush(); self._file.close()
      self._file = None
