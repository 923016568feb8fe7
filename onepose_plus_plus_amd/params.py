"""State-dict contract of the hot path.

`param_spec(cfg)` lists (key, shape, kind) in the exact order and with the exact names of the
reference `OnePosePlus_model.state_dict()` (195 entries for the shipped config):
  backbone.*            src/models/OnePosePlus/backbone/resnet.py:88-124 (ResNetFPN_8_2.__init__)
  kpt_3d_pos_encoding.* src/models/OnePosePlus/utils/position_encoding.py:49-79
  loftr_coarse/fine.*   src/models/OnePosePlus/loftr_module/transformer.py:7-63, :97-126
kind is one of: "conv", "bn_weight", "bn_bias", "bn_mean", "bn_var", "bn_count",
"linear_w", "linear_b", "xavier", "ln_weight", "ln_bias".
"""


def _bn(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_weight"),
        (prefix + ".bias", (c,), "bn_bias"),
        (prefix + ".running_mean", (c,), "bn_mean"),
        (prefix + ".running_var", (c,), "bn_var"),
        (prefix + ".num_batches_tracked", (), "bn_count"),
    ]


def _basic_block(prefix, cin, cout, stride):
    out = [
        (prefix + ".conv1.weight", (cout, cin, 3, 3), "conv"),
        (prefix + ".conv2.weight", (cout, cout, 3, 3), "conv"),
    ]
    out += _bn(prefix + ".bn1", cout)
    out += _bn(prefix + ".bn2", cout)
    if stride != 1:
        out.append((prefix + ".downsample.0.weight", (cout, cin, 1, 1), "conv"))
        out += _bn(prefix + ".downsample.1", cout)
    return out


def backbone_spec(cfg):
    r = cfg["loftr_backbone"]["resnetfpn"]
    d0 = r["initial_dim"]
    d1, d2, d3 = r["block_dims"]
    s = [("backbone.conv1.weight", (d0, 1, 7, 7), "conv")]
    s += _bn("backbone.bn1", d0)
    s += _basic_block("backbone.layer1.0", d0, d1, 1)
    s += _basic_block("backbone.layer1.1", d1, d1, 1)
    s += _basic_block("backbone.layer2.0", d1, d2, 2)
    s += _basic_block("backbone.layer2.1", d2, d2, 1)
    s += _basic_block("backbone.layer3.0", d2, d3, 2)
    s += _basic_block("backbone.layer3.1", d3, d3, 1)
    s.append(("backbone.layer3_outconv.weight", (d3, d3, 1, 1), "conv"))
    s.append(("backbone.layer2_outconv.weight", (d3, d2, 1, 1), "conv"))
    s.append(("backbone.layer2_outconv2.0.weight", (d3, d3, 3, 3), "conv"))
    s += _bn("backbone.layer2_outconv2.1", d3)
    s.append(("backbone.layer2_outconv2.3.weight", (d2, d3, 3, 3), "conv"))
    s.append(("backbone.layer1_outconv.weight", (d2, d1, 1, 1), "conv"))
    s.append(("backbone.layer1_outconv2.0.weight", (d2, d2, 3, 3), "conv"))
    s += _bn("backbone.layer1_outconv2.1", d2)
    s.append(("backbone.layer1_outconv2.3.weight", (d1, d2, 3, 3), "conv"))
    return s


def kpt_encoder_spec(cfg):
    k = cfg["keypoints_encoding"]
    chans = [3] + list(k["keypoints_encoder"]) + [k["descriptor_dim"]]
    s = []
    # nn.Sequential indices: Linear, InstanceNorm1d, ReLU triples -> 0, 3, 6, 9
    for i in range(1, len(chans)):
        idx = 3 * (i - 1)
        s.append(("kpt_3d_pos_encoding.encoder.%d.weight" % idx, (chans[i], chans[i - 1]), "linear_w"))
        s.append(("kpt_3d_pos_encoding.encoder.%d.bias" % idx, (chans[i],), "linear_b"))
    return s


def transformer_spec(name, tcfg):
    d = tcfg["d_model"]
    n_layers = len(list(tcfg["layer_names"])) * tcfg["layer_iter_n"]
    s = []
    for i in range(n_layers):
        p = "%s.layers.%d" % (name, i)
        s += [
            (p + ".q_proj.weight", (d, d), "xavier"),
            (p + ".k_proj.weight", (d, d), "xavier"),
            (p + ".v_proj.weight", (d, d), "xavier"),
            (p + ".merge.weight", (d, d), "xavier"),
            (p + ".mlp.0.weight", (2 * d, 2 * d), "xavier"),
            (p + ".mlp.2.weight", (d, 2 * d), "xavier"),
            (p + ".norm1.weight", (d,), "ln_weight"),
            (p + ".norm1.bias", (d,), "ln_bias"),
            (p + ".norm2.weight", (d,), "ln_weight"),
            (p + ".norm2.bias", (d,), "ln_bias"),
        ]
    if tcfg.get("final_proj"):      # constructed upstream but never applied (transformer.py:123-124):
        s += [(name + ".final_proj.weight", (d, d), "xavier"),      # kept so that a strict load of such a checkpoint works
              (name + ".final_proj.bias", (d,), "linear_b")]
    return s


def param_spec(cfg):
    s = backbone_spec(cfg)
    if cfg["keypoints_encoding"]["enable"]:
        s += kpt_encoder_spec(cfg)
    s += transformer_spec("loftr_coarse", cfg["loftr_coarse"])
    s += transformer_spec("loftr_fine", cfg["loftr_fine"])
    return s
