"""Architecture table of the spatial-action-map Q-network (host side, no torch).

Mirrors the reference's ``DataParallel(FCN).state_dict()`` naming so checkpoints
interchange (reference: networks.py:7-14, resnet.py:52-91, policies.py:39).
The device-side layout (flat NHWC/OHWI parameter buffer) comes from the C-ABI
(`simq_param_tensor_info`); this module only knows reference names and shapes.
"""

PREFIX = 'module.'
STATE_WIDTH = 96                       # envs.py:2010 Mapper.LOCAL_MAP_PIXEL_WIDTH
NUM_OUTPUT_CHANNELS = {                # envs.py:810 (pushing), envs.py:1090 (robots with hooks)
    'pushing_robot': 1, 'lifting_robot': 2, 'throwing_robot': 2, 'rescue_robot': 2,
}


def get_num_output_channels(robot_type):
    """envs.py:370-372."""
    if robot_type not in NUM_OUTPUT_CHANNELS:
        raise Exception(robot_type)
    return NUM_OUTPUT_CHANNELS[robot_type]


def get_action_space(robot_type):
    """envs.py:374-376."""
    return get_num_output_channels(robot_type) * STATE_WIDTH * STATE_WIDTH


def _bn(name, c):
    return [(name + '.weight', (c,), 'bn_weight'), (name + '.bias', (c,), 'bn_bias'),
            (name + '.running_mean', (c,), 'bn_mean'), (name + '.running_var', (c,), 'bn_var'),
            (name + '.num_batches_tracked', (), 'bn_count')]


def state_spec(num_input_channels, num_output_channels):
    """Ordered [(reference key, reference shape, kind)] -- 138 entries."""
    spec = []
    r = PREFIX + 'resnet18.'
    spec.append((r + 'conv1.weight', (64, num_input_channels, 7, 7), 'conv_w'))
    spec += _bn(r + 'bn1', 64)
    inplanes = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            b = '%slayer%d.%d.' % (r, li, bi)
            cin = inplanes if bi == 0 else planes
            spec.append((b + 'conv1.weight', (planes, cin, 3, 3), 'conv_w'))
            spec += _bn(b + 'bn1', planes)
            spec.append((b + 'conv2.weight', (planes, planes, 3, 3), 'conv_w'))
            spec += _bn(b + 'bn2', planes)
            if bi == 0 and cin != planes:
                spec.append((b + 'downsample.0.weight', (planes, cin, 1, 1), 'conv_w'))
                spec += _bn(b + 'downsample.1', planes)
        inplanes = planes
    spec.append((r + 'fc.weight', (1000, 512), 'fc_w'))
    spec.append((r + 'fc.bias', (1000,), 'fc_b'))
    p = PREFIX
    spec.append((p + 'conv1.weight', (128, 512, 1, 1), 'conv_w'))
    spec.append((p + 'conv1.bias', (128,), 'conv_b'))
    spec += _bn(p + 'bn1', 128)
    spec.append((p + 'conv2.weight', (32, 128, 1, 1), 'conv_w'))
    spec.append((p + 'conv2.bias', (32,), 'conv_b'))
    spec += _bn(p + 'bn2', 32)
    spec.append((p + 'conv3.weight', (num_output_channels, 32, 1, 1), 'conv_w'))
    spec.append((p + 'conv3.bias', (num_output_channels,), 'conv_b'))
    return spec


TRAINABLE_KINDS = ('conv_w', 'conv_b', 'bn_weight', 'bn_bias')   # tensors that receive a gradient
