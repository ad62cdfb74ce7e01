"""``python main.py <name> [--flags]``: the reference's dispatch contract (reference main.py:11-18)."""
from importlib import import_module

from animeface_amd.utils_argument import get_default_parser


def main():
    parser = get_default_parser()
    args = parser.parse_known_args()[0]
    module = import_module(f'.{args.name}', 'animeface_amd.implementations')
    module.main(parser)


if __name__ == '__main__':
    main()
