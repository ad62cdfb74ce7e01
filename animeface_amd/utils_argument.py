"""Argument helpers with the reference's flag names (reference utils/argument.py:10-63)."""
from argparse import ArgumentParser


def add_args(parser, arg_defaults, prefix='--'):
    for k, v in arg_defaults.items():
        option = prefix + k.replace('_', '-')
        default, help_ = (v[0], '') if len(v) == 1 else v
        value_type = type(default)
        if value_type in [float, int, str]:
            parser.add_argument(option, default=default, type=value_type, help=help_)
        elif value_type == bool:
            if default:
                raise Exception('Only supports store_true action')
            parser.add_argument(option, default=default, action='store_true', help=help_)
        elif value_type in [list, tuple]:
            parser.add_argument(option, default=default, type=type(default[0]), nargs='*', help=help_)
        elif isinstance(value_type, type):
            parser.add_argument(option, default=None, type=default, help=help_)
    return parser


def get_default_parser():
    parser = ArgumentParser()
    parser.add_argument('name')
    return add_args(parser, dict(
        image_size=[128, 'Size of image.'],
        batch_size=[32, 'Batch size'],
        dataset=['animeface', 'Dataset name'],
        min_year=[2005, 'Minimum of generated year. Ignored when dataset==danbooru'],
        num_images=[60000, 'Number of images to include in training set. Ignored when dataset==animeface'],
        save=[1000, 'Interval for saving the model'],
        max_iters=[-1, 'Maximum iterations to train the model. If < 0, it will be calculated using --default-epochs'],
        default_epochs=[100, 'Used to calculate the max iteration if --max-iters < 0'],
        disable_gpu=[False, 'Disable GPU'],
        disable_amp=[False, 'Disable AMP'],
        log_file=[str, 'Filename for saving log output'],
        log_interval=[1, 'Interval for logging to log file'],
        debug=[False, 'Debug mode']))
