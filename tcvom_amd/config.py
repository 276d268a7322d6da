"""Config tree with the reference's keys (config.py:3-44) without yacs: nested attribute dict, YAML merge and
`KEY.SUB value` command-line overrides (train_ddp.py:352-372)."""
import ast

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})

    def merge(self, other):
        for k, v in other.items():
            if k not in self:
                raise KeyError('unknown config key %s' % k)
            if isinstance(self[k], CfgNode):
                self[k].merge(v)
            else:
                self[k] = _coerce(v, self[k])

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, 'overrides come in KEY VALUE pairs'
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _coerce(val, node[parts[-1]])


def _coerce(v, like):
    if isinstance(v, str) and not isinstance(like, str):
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
    if isinstance(like, tuple) and isinstance(v, (list, str)):
        v = tuple(ast.literal_eval(v)) if isinstance(v, str) else tuple(v)
    if isinstance(like, float) and isinstance(v, (int, str)):
        v = float(v)
    return v


def get_cfg_defaults():
    c = CfgNode()
    c.MODEL = 'vmn50'
    c.AGG_WINDOW = 9
    c.SYSTEM = CfgNode(NUM_WORKERS=4, RANDOM_SEED=-1, OUTDIR='train_log', EXP_SUFFIX='', CUDNN_BENCHMARK=True,
                       CUDNN_DETERMINISTIC=False, CUDNN_ENABLED=True)
    c.DATASET = CfgNode(PATH='', SUBSET=False)
    c.TRAIN = CfgNode(LOAD_CKPT='', LOAD_OPT='', FREEZE_BACKBONE=False, BATCH_SIZE_PER_GPU=1, VAL_BATCH_SIZE_PER_GPU=1,
                      BASE_LR=5e-4, LR_STRATEGY='const', WEIGHT_DECAY=1e-4, TRAIN_INPUT_SIZE=(512, 512),
                      VAL_INPUT_SIZE=(512, 512), MIN_EDGE_LENGTH=1088, OPTIMIZER='adam', TOTAL_STEPS=50, PRINT_FREQ=10,
                      IMAGE_FREQ=500, VAL_START_EPOCH=15)       # VAL_START_EPOCH: train_ddp.py:323 hard-codes 15
    return c
