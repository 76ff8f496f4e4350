"""TEST-ONLY stand-in for absl.flags (see absl/__init__.py).

Semantics kept from real absl that matter for fixture fidelity:
  * DEFINE_float coerces its default with float(); DEFINE_integer with int().
  * FLAGS.name reads the value, FLAGS[name] returns the Flag holder (.value).
  * `name in FLAGS`, iteration over names, delattr(FLAGS, name), FLAGS(argv).
"""


class Flag(object):

  def __init__(self, name, default, help_string=""):
    self.name = name
    self.default = default
    self.value = default
    self.help = help_string


class FlagValues(object):

  def __init__(self):
    object.__setattr__(self, "_flags", {})

  # definition ---------------------------------------------------------------
  def _define(self, name, default, help_string=""):
    if name in self._flags:
      raise ValueError("The flag '%s' is defined twice." % name)
    self._flags[name] = Flag(name, default, help_string)

  # container protocol -------------------------------------------------------
  def __getattr__(self, name):
    flags_ = object.__getattribute__(self, "_flags")
    if name in flags_:
      return flags_[name].value
    raise AttributeError(name)

  def __setattr__(self, name, value):
    if name in self._flags:
      self._flags[name].value = value
    else:
      raise AttributeError(name)

  def __delattr__(self, name):
    if name in self._flags:
      del self._flags[name]
    else:
      raise AttributeError(name)

  def __getitem__(self, name):
    return self._flags[name]

  def __contains__(self, name):
    return name in self._flags

  def __iter__(self):
    return iter(list(self._flags))

  def __len__(self):
    return len(self._flags)

  def __call__(self, argv, known_only=False):
    return list(argv[:1])

  def is_parsed(self):
    return True

  def flag_values_dict(self):
    return {k: f.value for k, f in self._flags.items()}


FLAGS = FlagValues()


def DEFINE_bool(name, default, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, None if default is None else bool(default), help)


DEFINE_boolean = DEFINE_bool


def DEFINE_integer(name, default, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, None if default is None else int(default), help)


def DEFINE_float(name, default, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, None if default is None else float(default), help)


def DEFINE_string(name, default, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, default, help)


def DEFINE_enum(name, default, enum_values, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, default, help)


def DEFINE_list(name, default, help, flag_values=FLAGS, **kw):  # pylint: disable=redefined-builtin
  flag_values._define(name, default, help)


def register_validator(*a, **kw):
  pass


def mark_flag_as_required(*a, **kw):
  pass
