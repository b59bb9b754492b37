"""``python -m sai ...`` = ``python -m sai_amd ...`` (sai/__main__.py:64-76)."""

from sai_amd.__main__ import main

if __name__ == "__main__":
    main()
