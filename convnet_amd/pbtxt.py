"""Protobuf *text-format* reader and writer for the reference's model files (proto/convnet_config.proto).

There is no protoc / libprotobuf in the target image, and the hot path needs only the Model / Layer /
Edge / Optimizer messages, so this is a small schema-driven reader: unknown fields are an error
(like ``TextFormat::Parse``), unset optional fields read as their proto2 defaults, ``has_x()``
reports explicit presence, and ``MergeFrom`` follows proto2 semantics (used for the default
optimizers, src/convnet.cc:36-50).
"""
import re

# field -> default.  ``list`` marks repeated fields; a Msg subclass marks a sub-message.


class Msg:
    FIELDS = {}

    def __init__(self):
        object.__setattr__(self, "_set", {})

    def __getattr__(self, name):
        fields = type(self).FIELDS
        if name.startswith("has_") and name[4:] in fields:
            key = name[4:]
            return lambda: key in self._set
        if name not in fields:
            raise AttributeError(f"{type(self).__name__} has no field {name}")
        if name in self._set:
            return self._set[name]
        d = fields[name]
        if d is list:
            return self._set.setdefault(name, [])
        if isinstance(d, tuple) and d[0] is list:
            return self._set.setdefault(name, [])
        if isinstance(d, type) and issubclass(d, Msg):
            return d()  # unset sub-message reads as all-defaults (not stored)
        return d

    def __setattr__(self, name, value):
        if name not in type(self).FIELDS:
            raise AttributeError(f"{type(self).__name__} has no field {name}")
        self._set[name] = value

    def mutable(self, name):
        d = type(self).FIELDS[name]
        if name not in self._set:
            self._set[name] = d()
        return self._set[name]

    def MergeFrom(self, other):
        for k, v in other._set.items():
            d = type(self).FIELDS[k]
            if isinstance(v, list):
                self.__getattr__(k).extend(v)
            elif isinstance(d, type) and issubclass(d, Msg):
                self.mutable(k).MergeFrom(v)
            else:
                self._set[k] = v

    def CopyFrom(self, other):
        self._set.clear()
        self.MergeFrom(other)

    def copy(self):
        c = type(self)()
        c.MergeFrom(self)
        return c

    def __repr__(self):
        return f"{type(self).__name__}({self._set})"


class LayerSlice(Msg):
    FIELDS = {"name": "", "num_channels": 0}


class Optimizer(Msg):  # proto:63-113
    NONE, INVERSE_T, EXPONENTIAL, LINEAR, EXPONENTIAL_STEP = "NONE", "INVERSE_T", "EXPONENTIAL", "LINEAR", "EXPONENTIAL_STEP"
    FIELDS = {
        "optimizer_type": "STOCHASTIC_GRADIENT_DESCENT", "epsilon": 0.0, "epsilon_decay_timescale": 0,
        "initial_momentum": 0.0, "final_momentum": 0.0, "momentum_transition_timescale": 0, "l2_decay": 0.0,
        "weight_norm_limit": 0.0, "weight_norm_constraint": 0.0, "epsilon_decay": "NONE", "minimum_epsilon": 0.0,
        "decay_factor": 1.0, "gradient_clip": -1.0, "lbfgs_memory": 0, "start_optimization_after": 0,
        "adagrad_delta": 1.0, "rms_prop_factor": 0.0, "nesterov_momentum": False, "shared_prior": False,
        "shared_prior_cost": 0.0, "shared_prior_file": "",
    }


class Layer(Msg):  # proto:12-61
    FIELDS = {
        "name": "", "num_channels": 0, "size": -1, "dropprob": 0.0, "is_input": False, "activation": "LINEAR",
        "image_size_y": 1, "image_size_x": 1, "display": False, "is_output": False, "gaussian_dropout": False,
        "max_act_gaussian_dropout": -1.0, "gpu_id": 0, "hinge_margin": 0.0, "layer_slice": (list, LayerSlice),
        "loss_function": "CROSS_ENTROPY_MULTINOMIAL", "performance_metric": "CLASSIFICATION_MULTINOMIAL",
        "loss_function_weight": 1.0, "tied_data": "", "image_size_t": 1, "batch_normalize": False, "bn_f": 0.98,
        "bn_epsilon": 1e-5, "gamma_optimizer": Optimizer, "beta_optimizer": Optimizer,
    }


class Edge(Msg):  # proto:115-222
    FIELDS = {
        "source": "", "dest": "", "edge_type": "FC", "kernel_size": -1, "stride": 1, "padding": 0,
        "initialization": "DENSE_GAUSSIAN_SQRT_FAN_IN", "init_wt": 1.0, "init_bias": 0.0,
        "weight_optimizer": Optimizer, "bias_optimizer": Optimizer, "shared_bias": False, "block_backprop": False,
        "tied_to": "", "has_no_bias": False, "scale_gradients": 1.0, "partial_sum": 0, "sample_factor": 1,
        "response_norm_in_blocks": False, "add_scale": 0.0, "pow_scale": 0.0, "frac_of_filters_response_norm": 0.0,
        "gpu_id": 0, "pretrained_model": "", "pretrained_edge_name": "", "display": False, "source_slice": "",
        "dest_slice": "", "grad_check": False, "grad_check_num_params": 0, "grad_check_epsilon": list,
        "kernel_size_y": 0, "kernel_size_x": 0, "kernel_size_t": 0, "stride_y": 1, "stride_x": 1, "stride_t": 1,
        "padding_y": 0, "padding_x": 0, "padding_t": 0,
    }


class Model(Msg):  # proto:239-272
    FIELDS = {
        "name": "", "layer": (list, Layer), "edge": (list, Edge), "seed": 0, "max_iter": -1, "display_after": -1,
        "save_after": -1, "image_size": 0, "patch_size": 0, "print_after": -1, "localizer": False,
        "checkpoint_dir": "", "print_weights": False, "timestamp": list, "display": False, "validate_after": -1,
        "reduce_lr_factor": 1.0, "reduce_lr_num_steps": 0, "reduce_lr_max": 0, "reduce_lr_threshold": 0.0,
        "default_weight_optimizer": Optimizer, "default_bias_optimizer": Optimizer, "polyak_queue_size": 0,
        "smaller_is_better": False, "polyak_after": 0, "reduce_lr_layer_name": "",
    }


_TOKEN = re.compile(r'\s*(?:#[^\n]*\n\s*)*("(?:[^"\\]|\\.)*"|[{}:<>]|[^\s{}:<>"#]+)')


def _tokens(text):
    pos, n = 0, len(text)
    text = text + "\n"
    while True:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() and not text[pos:].strip().startswith("#"):
                raise ValueError(f"pbtxt: cannot tokenise near {text[pos:pos + 40]!r}")
            return
        yield m.group(1)
        pos = m.end()
        if pos >= n:
            return


def _scalar(tok, default):
    if tok.startswith('"'):
        return bytes(tok[1:-1], "utf-8").decode("unicode_escape")
    if isinstance(default, bool):
        if tok in ("true", "True", "1"):
            return True
        if tok in ("false", "False", "0"):
            return False
        raise ValueError(f"pbtxt: bad bool {tok}")
    if isinstance(default, int):
        return int(tok)
    if isinstance(default, float):
        return float(tok)
    return tok  # enum identifier or repeated scalar decided by caller


def _parse_into(msg, toks, closer):
    fields = type(msg).FIELDS
    for tok in toks:
        if tok == closer:
            return
        if tok not in fields:
            raise ValueError(f"pbtxt: {type(msg).__name__} has no field named {tok!r}")
        name, d = tok, fields[tok]
        nxt = next(toks)
        if nxt == ":":
            nxt = next(toks)
        sub = d[1] if isinstance(d, tuple) else d
        if nxt in ("{", "<"):
            if not (isinstance(sub, type) and issubclass(sub, Msg)):
                raise ValueError(f"pbtxt: field {name} is not a message")
            child = sub()
            _parse_into(child, toks, "}" if nxt == "{" else ">")
            if isinstance(d, tuple):
                getattr(msg, name).append(child)
            else:
                msg.mutable(name).MergeFrom(child)
        elif d is list:
            v = nxt
            try:
                v = float(nxt) if ("." in nxt or "e" in nxt.lower()) else int(nxt)
            except ValueError:
                v = _scalar(nxt, "")
            getattr(msg, name).append(v)
        else:
            msg._set[name] = _scalar(nxt, d)
    if closer is not None:
        raise ValueError("pbtxt: unexpected end of input")


def parse(text, cls=Model):
    msg = cls()
    _parse_into(msg, iter(_tokens(text)), None)
    return msg


def read(path, cls=Model):
    """ReadPbtxt<T> (src/util.cc:87-102)."""
    with open(path) as f:
        return parse(f.read(), cls)


def _fmt(v, default):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        # enum identifiers are written bare, strings quoted: an enum field's default is one of its upper-case identifiers
        if isinstance(default, str) and default and default.isupper():
            return v
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, float):
        return repr(v)
    return str(v)


def dump(msg, indent=0):
    """Text format of the explicitly set fields, in schema order (what ``TextFormat::Print`` writes): ``parse(dump(m))`` rebuilds m."""
    pad, out = "  " * indent, []
    for name, d in type(msg).FIELDS.items():
        if name not in msg._set:
            continue
        v = msg._set[name]
        sub = d[1] if isinstance(d, tuple) else d
        if isinstance(sub, type) and issubclass(sub, Msg):
            for child in (v if isinstance(v, list) else [v]):
                out.append(f"{pad}{name} {{\n{dump(child, indent + 1)}{pad}}}\n")
        elif isinstance(v, list):
            for x in v:
                out.append(f"{pad}{name}: {_fmt(x, '')}\n")
        else:
            out.append(f"{pad}{name}: {_fmt(v, d)}\n")
    return "".join(out)


def write(path, msg):
    """WritePbtxt<T> (src/util.cc:104-112)."""
    with open(path, "w") as f:
        f.write(dump(msg))
