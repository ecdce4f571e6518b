#!/usr/bin/env python3
"""lrbinner.py {reads,contigs} -- the reference's command line (lrbinner.py:12-203),
flag for flag, in front of the MI355X-native hot path (lrbinner_amd)."""
import argparse
import logging
import os
import shutil
import sys
import time

from lrbinner_amd import pipelines

VERSION = "2.1"
DESCRIPTION = ("LRBinner Help. A tool developed for binning of metagenomics long reads (PacBio/ONT) "
               "and long read assemblies. Tool utilizes composition and coverage profiles of reads "
               "based on k-mer frequencies to perform dimension reduction via a deep variational "
               "auto-encoder. Dimension reduced reads are then clustered. Minimum RAM requirement "
               "is 9GB (4GB GPU if cuda used).")

ENVIRONMENT = """environment (this build; the full list is in INTEGRATION.md):
  LRB_GPUS=N                   run the three profile stages on N GPUs of this node (reads sharded, one RCCL
                               all-reduce of the 15-mer table); default 1
  LRB_COLLECTIVE_TIMEOUT_S=S   how long a rank waits in a collective of that job before the job fails
                               (default 600; the first one also absorbs the ranks' parse skew -- raise it for
                               very large inputs on slow file systems)
  LRB_DEVICE=I                 GPU of a single-GPU run (default 0)
  LRB_SEED=S                   seed random / numpy / torch (the reference never seeds)
  LRB_VAE_DETERMINISTIC=1      with LRB_SEED: the VAE's sums in a fixed order -- the same latents run after run
"""


def build_parser():
    common = argparse.ArgumentParser(add_help=False)
    add = common.add_argument
    add('--reads-path', '-r', type=str, required=True, help="Reads path for binning")
    add('--k-size', '-k', type=int, choices=[3, 4, 5], default=3,
        help="k value for k-mer frequency vector. Choose between 3 and 5.")
    add('--bin-size', '-bs', type=int, default=10, help="Bin size for the coverage histogram.")
    add('--bin-count', '-bc', type=int, default=32,
        help="Number of bins for the coverage histogram.")
    add('--ae-epochs', type=int, default=200, help="Epochs for the auto_encoder.")
    add('--ae-dims', type=int, default=8, help="Size of the latent dimension.")
    add('--ae-hidden', type=str, default="128,128", help="Hidden layer sizes eg: 128,128")
    add('--threads', '-t', type=int, default=8, help="Thread count for computations")
    add('--separate', '-sep', action='store_true',
        help="Flag to separate reads/contigs into bins detected. Avaialbe in folder named 'binned'.")
    add('--cuda', action='store_true', help="Whether to use the GPU for the VAE if available.")
    add('--resume', action='store_true',
        help="Continue from the last step or the binning step (which ever comes first). "
             "Can save time needed count k-mers.")
    add('--output', '-o', metavar='<DEST>', type=str, required=True, help="Output directory")

    main = argparse.ArgumentParser(description=DESCRIPTION, add_help=True, epilog=ENVIRONMENT,
                                   formatter_class=argparse.RawDescriptionHelpFormatter)
    main.add_argument('--version', '-v', action='version', help="Show version.",
                      version=f'%(prog)s {VERSION}')
    modes = main.add_subparsers(title="LRBinner running Mode", required=True, dest="mode")
    reads = modes.add_parser('reads', parents=[common], help="for binning reads")
    reads.add_argument('--min-bin-size', '-mbs', type=int, default=10000,
                       help="The minimum number of reads a bin should have.")
    reads.add_argument('--bin-iterations', '-bit', type=int, default=1000,
                       help="Number of iterations for cluster search. Use 0 for exhaustive search.")
    contigs = modes.add_parser('contigs', parents=[common], help="for binning contigs")
    contigs.add_argument('--contigs', '-c', type=str, required=True, help="Contigs path")
    return main


def main(argv=None):
    args = build_parser().parse_args(argv)
    output = args.output

    logger = logging.getLogger('LRBinner')
    logger.setLevel(logging.DEBUG)
    # one rank of `python -m torch.distributed.run ... lrbinner.py reads`: ranks other than 0 only take their share
    # of the profile stages (lrbinner_amd.pipelines.run_profile_rank); the output directory is rank 0's to prepare
    from lrbinner_amd.dist import launcher_world
    rank, world, _ = launcher_world()
    if world > 1 and rank > 0:
        if args.mode == 'reads' and os.path.isfile(args.reads_path):
            pipelines.run_profile_rank(args)
        return
    if not args.resume and os.path.isdir(output):
        shutil.rmtree(output)
    formatter = logging.Formatter('%(asctime)s - %(levelname)s - %(message)s')
    console = logging.StreamHandler()
    console.setFormatter(formatter)
    console.setLevel(logging.INFO)
    logger.addHandler(console)

    if args.reads_path.split(".")[-1].lower() not in ['fq', 'fasta', 'fa', 'fastq']:
        logger.error("Unable to detect file type of reads. Please use either FASTA of FASTQ. Good Bye!")
        sys.exit(1)
    # limits of the HIP kernels behind the stages, checked before any work starts (the reference
    # takes any value; its kernels are loops): K4 holds a latent row in 64 registers, K3 a
    # histogram of at most 1024 bins per wave in LDS, K1 covers k = 3..5 as the reference's help says
    if not 1 <= args.ae_dims <= 64:
        logger.error("--ae-dims must be between 1 and 64 in this build. Good Bye!")
        sys.exit(1)
    if not 1 <= args.bin_count <= 1024 or args.bin_size < 1:
        logger.error("--bin-count must be between 1 and 1024 and --bin-size at least 1 in this build. Good Bye!")
        sys.exit(1)
    if not 3 <= args.k_size <= 5:
        logger.error("--k-size must be 3, 4 or 5. Good Bye!")
        sys.exit(1)
    if args.threads <= 0:
        print("Minimum number of threads is 1. Using thread count 1 and continue")
        args.threads = 1
    if not os.path.isfile(args.reads_path):
        print("Failed to open reads file")
        print("Exitting process. Good Bye!")
        sys.exit(1)

    if not os.path.exists(output):
        os.makedirs(f"{output}/profiles")
    if args.mode == 'contigs':
        os.makedirs(f"{output}/marker_genes", exist_ok=True)
        os.makedirs(f"{output}/fragments", exist_ok=True)

    logfile = logging.FileHandler(f"{output}/LRBinner.log")
    logfile.setLevel(logging.DEBUG)
    logfile.setFormatter(formatter)
    logger.addHandler(logfile)

    start = time.time()
    logger.info("Command " + " ".join(sys.argv))
    if os.environ.get("LRB_SEED"):
        # the reference never seeds anything (its runs differ from one to the next); for repeatable
        # runs of this build LRB_SEED seeds the three generators the stages draw from
        import random
        import numpy
        import torch
        seed = int(os.environ["LRB_SEED"])
        random.seed(seed)
        numpy.random.seed(seed % (2 ** 32))
        torch.manual_seed(seed)
        logger.info(f"LRB_SEED {seed}")
    if args.cuda:
        import torch
        # (device_count does not initialise the GPU: the multi-GPU profile stages start their ranks first)
        if torch.cuda.device_count() > 0:
            logger.info("CUDA found in system")
        else:
            # unlike the reference (lrbinner.py:175-182 downgrades a local only) the
            # downgrade reaches the VAE stage
            args.cuda = False
            logger.info("CUDA not found in system")

    if args.mode == 'contigs':
        pipelines.run_contig_binning(args)
    else:
        pipelines.run_reads_binning(args)

    logger.info("Program Finished!. Please find the output in bins.txt")
    logger.info(f"Total time consumed = {time.time() - start:.2f} seconds")
    logger.info("Thank you for using LRBinner. Feedback will be much appreciated!")
    logger.removeHandler(logfile)
    logger.removeHandler(console)


if __name__ == '__main__':
    main()
