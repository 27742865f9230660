#!/bin/bash
set -e
# set -o pipefail

############################################################
#  Program: speedseq
#  Version: 0.1.2
#  Author: Colby Chiang (cc2qe@virginia.edu)
############################################################

# force bitwise sorting
LC_ALL=C

# source the paths to the binaries used in the script
function source_binaries() {
    if [[ -e $1 ]]
    then
	echo "Sourcing executables from $1 ..."
	if [[ $1 == /* ]]
	then
	    source $1
	else
	    source ./$1
	fi
    else
	echo "Config file $1 not found. Attempting to auto-source executables"
	# general
	SPEEDSEQ_HOME=$( dirname `which speedseq` )
	SAMBAMBA=`which sambamba || true`
	SAMTOOLS=`which samtools || true`
	BGZIP=`which bgzip || true`
	TABIX=`which tabix || true`
	VAWK=`which vawk || true`
	PARALLEL=`which parallel || true`
	PYTHON=`which python2.7 || true`
    HEXDUMP=`which hexdump || true`

	# align
	BWA=`which bwa || true`
	SAMBLASTER=`which samblaster || true`

	# var/somatic
	FREEBAYES=`which freebayes || true`
	VEP=`which variant_effect_predictor.pl || true`
	VEP_CACHE_DIR=$SPEEDSEQ_HOME/annotations/vep_cache

	# sv
	LUMPYEXPRESS=`which lumpyexpress || true`
	LUMPY=`which lumpy || true`
	PAIREND_DISTRO=`which pairend_distro.py || true`
	SVTYPER=`which svtyper || true`
	BAMGROUPREADS=`which bamgroupreads.py || true`

        # CNVnator
	CNVNATOR_WRAPPER=`which cnvnator_wrapper.py || true`
	CNVNATOR=`which cnvnator || true`
	ANNOTATE_RD=`which annotate_rd.py || true`
	CNVNATOR_CHROMS_DIR=~/genomes/GRCh37/chroms

	# realign
	BAMTOFASTQ=`which bamtofastq.py || true`
	MBUFFER=`which mbuffer || true`
	BAMHEADRG=`which bamheadrg.py || true`
	BAMCLEANHEADER=`which bamcleanheader.py || true`
    fi
}

# ensure that the require python modules are installed before
# beginning analysis
function check_python_modules() {
    PYTHON_TEST=$1
    echo -e "\nChecking for required python modules ($PYTHON_TEST)..."

    PYSAM=`$PYTHON_TEST -c "import imp; imp.find_module('pysam')" 2>&1 || true`
    NUMPY=`$PYTHON_TEST -c "import imp; imp.find_module('numpy')" 2>&1 || true`
    SCIPY=`$PYTHON_TEST -c "import imp; imp.find_module('scipy')" 2>&1 || true`

    if [[ ! -z "$PYSAM" ]]
    then
	usage
	echo  -e "Error: pysam is not installed for $PYTHON_TEST\n"
	exit 1
    elif [[ ! -z "$NUMPY" ]]
    then
	usage
	echo  -e "Error: numpy is not installed for $PYTHON_TEST\n"
	exit 1
    elif [[ ! -z "$SCIPY" ]]
    then
	usage
	echo  -e "Error: scipy is not installed for $PYTHON_TEST\n"
	exit 1
    fi
}

# Check if a file is a cram or bam based on the binary header
function is_cram () {
    local file_path=$1

    header=$($HEXDUMP -n 4 -e '"%c"' ${file_path})
    if [ ${header} == 'CRAM' ]; then
        result=0
    else
        result=1
    fi

    return ${result}
}

function is_bam () {
    local file_path=$1
    local result=1

    is_gzipped=$($HEXDUMP -n 2 -e '"%x"' ${file_path})
    if [ ${is_gzipped} = '8b1f' ]; then
        is_bam=$(zcat ${file_path} | $HEXDUMP -n 3 -e '"%c"')
        if [ ${is_bam} = 'BAM' ]; then
            result=0
        fi
    fi

    return ${result}
}

# ensure that we have a valid CRAM or BAM file
function check_valid_cram_or_bam () {
    local input_path=$1
    if is_cram ${input_path} || is_bam ${input_path} ; then
        echo -e "${input_path} is a valid CRAM/BAM file"
    else
        echo -e "${input_path}: is NOT a valid CRAM/BAM file!"
        exit 1
    fi
}

## global usage
function usage() {
    echo "
Program: speedseq
Version: 0.1.2
Author: Colby Chiang (cc2qe@virginia.edu)

usage:   speedseq <command> [options]

command: align    align FASTQ files with BWA-MEM
         var      call SNV and indel variants with FreeBayes
         somatic  call somatic SNV and indel variants in a tumor/normal pair with FreeBayes
         sv       call SVs with LUMPY
         realign  realign from a coordinate sorted BAM file

options: -h       show this message
"
}

function somatic_filter() {
    awk -v MINQUAL="$1" -v SSC_THRES="$2" -v ONLY_SOMATIC="$3" 'BEGIN {NORMAL=10; TUMOR=11; GL_IDX=0;}
    {
        if ($0~"^#") { print ; next; }
        if (! GL_IDX) {
            split($9,fmt,":")
            for (i=1;i<=length(fmt);++i) { if (fmt[i]=="GL") GL_IDX=i }
        }
        split($NORMAL,N,":");
        split(N[GL_IDX],NGL,",");
        split($TUMOR,T,":");
        split(T[GL_IDX],TGL,",");
        LOD_NORM=NGL[1]-NGL[2];
        LOD_TUMOR_HET=TGL[2]-TGL[1];
        LOD_TUMOR_HOM=TGL[3]-TGL[1];

        if (LOD_TUMOR_HET > LOD_TUMOR_HOM) { LOD_TUMOR=LOD_TUMOR_HET }
        else { LOD_TUMOR=LOD_TUMOR_HOM }

        DQUAL=LOD_TUMOR+LOD_NORM;

        if (DQUAL>=SSC_THRES && $NORMAL~"^0/0") {
            $7="PASS"
            $8="SSC="DQUAL";"$8
            print
        }
        else if (!ONLY_SOMATIC && $6>=MINQUAL && $10~"^0/0" && ! match($11,"^0/0")) {
            $8="SSC="DQUAL";"$8
            print
        }
    }' OFS="\t"
}

# alignment with BWA-MEM
function align() {
    function align_usage() {
	echo "
usage:   speedseq align [options] <reference.fa> <in1.fq> [in2.fq]

positional args:
         reference.fa
                  fasta file (indexed with bwa)
         in1.fq   paired-end fastq file. if -p flag is used then expected to be
                    an interleaved paired-end fastq file, and in2.fq may be omitted.
                    (can be gzipped)
         in2.fq   paired-end fastq file. (can be gzipped)

alignment options:
         -o STR   output prefix [in1.fq]
         -R STR   read group header line such as \"@RG\tID:id\tSM:samplename\tLB:lib\" (required)
         -p       first fastq file consists of interleaved paired-end sequences
         -t INT   threads [1]
         -T DIR   temp directory [./output_prefix.XXXXXXXXXXXX]
         -I FLOAT[,FLOAT[,INT[,INT]]]
                  specify the mean, standard deviation (10% of the mean if absent), max
                    (4 sigma from the mean if absent) and min of the insert size distribution.
                    FR orientation only. [inferred]

samblaster options:
         -i       include duplicates in splitters and discordants
         -c INT   maximum number of split alignments for a read to be included in splitter file [2]
         -m INT   minimum non-overlapping base pairs between two alignments for a read to be included in splitter file [20]

sambamba options:
         -M       amount of memory in GB to be used for sorting [20]

global options:
         -K FILE  path to speedseq.config file (default: same directory as speedseq)
         -v       verbose
         -h       show this message
"
    }

    # Check options passed in.
    if test -z "$2"
    then
	align_usage
	exit 1
    fi

    # set defaults
    SPEEDSEQ_DIR=`dirname $0`
    CONFIG="$SPEEDSEQ_DIR/speedseq.config"
    INTERLEAVED=0
    RG_FMT=""
    OUTPUT=""
    INCLUDE_DUPS="--excludeDups"
    MAX_SPLIT_COUNT=2
    MIN_NON_OVERLAP=20
    THREADS=1
    TEMP_DIR=""
    VERBOSE=1
    INS_DIST=""
    SORT_MEM=20 # amount of memory for sorting, in gigabytes

    while getopts ":hw:o:R:pic:m:M:t:T:I:vK:" OPTION
    do
	case "${OPTION}" in
	    h)
		align_usage
		exit 1
		;;
	    R)
		RG="$OPTARG"
		RG_FMT="-R '$OPTARG'"
		;;
	    p)
		INTERLEAVED=1
		;;
	    o)
		OUTPUT="$OPTARG"
		;;
	    i)
		INCLUDE_DUPS=""
		;;
	    c)
		MAX_SPLIT_COUNT="$OPTARG"
		;;
	    m)
		MIN_NON_OVERLAP="$OPTARG"
		;;
	    M)
		SORT_MEM="$OPTARG"
		;;
	    t)
		THREADS="$OPTARG"
		;;
	    T)
		TEMP_DIR="$OPTARG"
		;;
	    I)
		INS_DIST="-I $OPTARG"
		;;
	    v)
		VERBOSE=1
		;;
	    K)
		CONFIG="$OPTARG"
		;;
	esac
    done

    if [[ "$INTERLEAVED" -eq 1 ]]
    then
	REF="${@:${OPTIND}:1}"
	FQ="${@:$((${OPTIND}+1)):1}"
	if [[ -z "$OUTPUT" ]]
	then
	    OUTPUT=`basename "$FQ"`
	fi

	if [[ -z "$FQ" ]]
	then
            align_usage
            echo -e "Error: Fastq file $FQ not found.\n"
            exit 1
	fi
    else
	REF="${@:${OPTIND}:1}"
	FQ1="${@:$((${OPTIND}+1)):1}"
	FQ2="${@:$((${OPTIND}+2)):1}"
	if [[ -z "$OUTPUT" ]]
	then
	    OUTPUT=`basename "$FQ1"`
	fi

	if [[ -z "$FQ1" ]]
	then
            align_usage
            echo -e "Error: Fastq file $FQ1 not found.\n"
            exit 1
	elif [[ -z "$FQ2" ]]
	then
            align_usage
            echo -e "Error: Fastq file $FQ2 not found. (single-end reads not supported, use -p for interleaved FASTQ)\n"
            exit 1
	fi
    fi

    # Check that the ref and fastq files exist
    if [[ -z "$REF" ]] || [[ ! -f "$REF" ]]
    then
	align_usage
	echo -e "Error: Reference file $REF not found.\n"
	exit 1
    fi

    OUTBASE=`basename "$OUTPUT"`

    # Check for readgroup flag
    if [[ -z $RG_FMT ]]
    then
	align_usage
	echo -e "Error: no readgroup found. Please set a readgroup with the -R flag.\n"
	exit 1
    fi

    # Check that SORT_MEM > 2
    if [[ "$SORT_MEM" -lt 3 ]]
    then
	align_usage
	echo -e "Error: -M must be greater than 2"
	exit 1
    fi

    # Check the for the relevant binaries
    source_binaries $CONFIG

    if [[ ! -f "$BWA" ]]
    then
	align_usage
        echo -e "Error: bwa executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f  "$SAMBLASTER" ]]
    then
	align_usage
        echo -e "Error: samblaster executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$SAMBAMBA" ]]
    then
	align_usage
        echo -e "Error: sambamba executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$PARALLEL" ]]
    then
	align_usage
        echo -e "Error: parallel executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    fi

    # Check for BWA index of the reference
    if [[ ! -f "$REF.bwt" ]] || [[ ! -f "$REF.pac" ]] || [[ ! -f "$REF.ann" ]] || [[ ! -f "$REF.amb" ]] || [[ ! -f "$REF.sa" ]]
    then
	echo "Warning: Reference file not indexed with BWA. Indexing now..."
	$BWA index $REF
	echo "Done"
    fi

    echo "Aligning..."
    # create temp directory if not specified by command argument
    if [[ -z $TEMP_DIR ]]
    then
	TEMP_DIR=`mktemp -d ${OUTBASE}.XXXXXXXXXXXX`
    fi

    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        mkdir -p $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
        mkfifo $TEMP_DIR/spl_pipe $TEMP_DIR/disc_pipe"
    fi

    # create temp files
    mkdir -p $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
    if [[ ! -e $TEMP_DIR/spl_pipe ]]
    then
	mkfifo $TEMP_DIR/spl_pipe
    fi
    if [[ ! -e $TEMP_DIR/disc_pipe ]]
    then
	mkfifo $TEMP_DIR/disc_pipe
    fi

    # alignment command
    if [[ "$INTERLEAVED" -eq 1 ]]
    then
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        $BWA mem -t $THREADS -p $INS_DIST $RG_FMT $REF $FQ | \\
            $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe | \\
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \\
            $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $OUTPUT.bam /dev/stdin

        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe | \\
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \\
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $OUTPUT.splitters.bam /dev/stdin
        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe | \\
            $SAMBAMBA view -S -f bam /dev/stdin | \\
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $OUTPUT.discordants.bam /dev/stdin"
	fi

	echo "
        $BWA mem -t $THREADS -p $INS_DIST $RG_FMT $REF $FQ | \
	    $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe | \
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \
	    $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $OUTPUT.bam /dev/stdin

        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe | \
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \
	    $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $OUTPUT.splitters.bam /dev/stdin
        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe | \
            $SAMBAMBA view -S -f bam /dev/stdin | \
	    $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $OUTPUT.discordants.bam /dev/stdin
        " | $PARALLEL -j 3
    else
	if [[ $VERBOSE -eq 1 ]]
	then
        echo -e "
        $BWA mem -t $THREADS $INS_DIST $RG_FMT $REF $FQ1 $FQ2 | \\
            $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe | \\
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \\
            $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $OUTPUT.bam /dev/stdin

        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe | \\
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \\
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $OUTPUT.splitters.bam /dev/stdin
        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe | \\
            $SAMBAMBA view -S -f bam /dev/stdin | \\
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $OUTPUT.discordants.bam /dev/stdin"
	fi

        echo "
        $BWA mem -t $THREADS $INS_DIST $RG_FMT $REF $FQ1 $FQ2 | \
            $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe | \
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \
            $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $OUTPUT.bam /dev/stdin

        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe | \
            $SAMBAMBA view -S -f bam -l 0 /dev/stdin | \
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $OUTPUT.splitters.bam /dev/stdin
        gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe | \
            $SAMBAMBA view -S -f bam /dev/stdin | \
            $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $OUTPUT.discordants.bam /dev/stdin
        " | $PARALLEL -j 3
    fi
    
    # index the files
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        $SAMBAMBA index $OUTPUT.bam
            $SAMBAMBA index $OUTPUT.discordants.bam
            $SAMBAMBA index $OUTPUT.splitters.bam"
    fi

    echo "
    $SAMBAMBA index $OUTPUT.bam
    $SAMBAMBA index $OUTPUT.discordants.bam
    $SAMBAMBA index $OUTPUT.splitters.bam
    " | $PARALLEL -j 3

    # clean up
    rm -r $TEMP_DIR

    echo "Done"

    # exit cleanly
    exit 0
}

function var() {
    function var_usage() {
        echo "
usage:   speedseq var [options] <reference.fa> <input1.bam> [input2.bam [...]]

positional args:
         reference.fa
                  genome reference fasta file
         input.bam
                  BAM file(s) to call variants on. Must have readgroup information,
                    and the SM readgroup tags will be the VCF column header

options:
         -o STR   output prefix [input1.bam]
         -w FILE  BED file of windowed genomic intervals
         -q FLOAT minimum variant QUAL score to output [1]
         -t INT   threads [1]
         -T DIR   temp directory [./output_prefix.XXXXXXXXXXXX]
         -A       annotate the vcf with VEP
         -a       VEP assembly to use [GRCh37]
         -K FILE  path to speedseq.config file (default: same directory as speedseq)
         -v       verbose
         -k       keep temporary files
         -h       show this message
"
    }

    # Check options passed in.
    if test -z "$2"
    then
        var_usage
        exit 1
    fi

    # set defaults
    SPEEDSEQ_DIR=`dirname $0`
    CONFIG="$SPEEDSEQ_DIR/speedseq.config"
    THREADS=1
    MINQUAL=1
    TEMP_DIR=""
    ANNOTATE=0
    MINQUAL=1
    VERBOSE=1
    KEEP=0
    VEP_ASSEMBLY="GRCh37"

    while getopts ":ho:w:t:T:Aa:q:vkK:" OPTION
    do
        case "${OPTION}" in
            h)
                var_usage
                exit 1
                ;;
            o)
                OUTPUT="$OPTARG"
                ;;
	    w)
		WINDOWS="$OPTARG"
		;;
	    q)
		MINQUAL="$OPTARG"
		;;
            t)
                THREADS="$OPTARG"
                ;;
            T)
                TEMP_DIR="$OPTARG"
                ;;
	    A)
		ANNOTATE=1
		;;
	    a)
		VEP_ASSEMBLY="$OPTARG"
		;;
            v)
                VERBOSE=1
                ;;
	    k)
		KEEP=1
		;;
	    K)
		CONFIG="$OPTARG"
		;;
        esac
    done

    # parse the positional arguments
    REF="${@:${OPTIND}:1}"
    BAM_STRING="${@:$((${OPTIND}+1))}"
    BAM_LIST=($BAM_STRING)
    if [[ -z $OUTPUT ]]
    then
	OUTPUT=`basename "${BAM_LIST[0]}"`
    fi
    OUTBASE=`basename "$OUTPUT"`

    OPTIND=0

    # Check the for the relevant binaries
    source_binaries $CONFIG

    if [[ ! -f "$FREEBAYES" ]]
    then
	var_usage
        echo -e "Error: freebayes executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$BGZIP" ]]
    then
	var_usage
        echo -e "Error: bgzip executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$TABIX" ]]
    then
	var_usage
        echo -e "Error: tabix executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$VEP" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
	var_usage
        echo -e "Error: VEP not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -d "$VEP_CACHE_DIR" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
	var_usage
        echo -e "Error: VEP cache directory not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$PARALLEL" ]]
    then
	var_usage
        echo -e "Error: parallel executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$VAWK" ]]
    then
	var_usage
        echo -e "Error: vawk executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    fi

    # Check that the ref and bam files exist
    if [[ -z "$REF" ]] || [[ ! -f "$REF" ]]
    then
	var_usage
	echo -e "Error: Reference file $REF not found.\n"
	exit 1
    fi

    for TEST_BAM in ${BAM_LIST[@]}
    do
	if [[ ! -f $TEST_BAM ]]
	then
	    var_usage
	    echo -e "Error: BAM file $TEST_BAM not found.\n"
	    exit 1
	fi
    done

    echo "Calling variants..."
    # make temporary directory
    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        create temporary directory"
    fi
    if [[ -z $TEMP_DIR ]]
    then
	TEMP_DIR=`mktemp -d ${OUTBASE}.XXXXXXXXXXXX`
    else
	mkdir -p $TEMP_DIR	
    fi

    # if no windows file, then make naive windows based on the chroms
    if [[ -z $WINDOWS ]]
    then
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        $SAMBAMBA view -H ${BAM_LIST[0]} | grep \"^@SQ\" | cut -f 2- | awk '{ gsub(\"^SN:\",\"\",\$1); gsub(\"^LN:\",\"\",\$2); print \$1\"\\\t0\\\t\"\$2; }' > $TEMP_DIR/windows.bed"
	fi

	$SAMBAMBA view -H ${BAM_LIST[0]} | grep "^@SQ" | cut -f 2- | awk '{ gsub("^SN:","",$1); gsub("^LN:","",$2); print $1"\t0\t"$2; }' > $TEMP_DIR/windows.bed
	WINDOWS="$TEMP_DIR/windows.bed"
    fi
    
    # construct the parallel command over windows
    if [[ $VERBOSE -eq 1 ]]
    then
        echo -e "
        $FREEBAYES \\
            -f $REF \\
            --region \$chrom:\$start..\$end \\
            $BAM_STRING \\
            --min-repeat-entropy 1 \\
            | $VAWK --header '\$6>=$MINQUAL && I\$RPR>0 && I\$RPL>0' \\
            > ${TEMP_DIR}/$OUTBASE.\$i.vcf"
    fi
    
    for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
    do
        echo "$FREEBAYES \
        -f $REF \
        --region $i \
        --min-repeat-entropy 1 \
        $BAM_STRING \
        | $VAWK --header '\$6>=$MINQUAL && I\$RPR>0 && I\$RPL>0' \
        > ${TEMP_DIR}/$OUTBASE.$i.vcf"
    done > $TEMP_DIR/var_command.txt

    # run the parallel freebayes command
    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        cat $TEMP_DIR/var_command.txt | $PARALLEL -j $THREADS"
    fi
    cat $TEMP_DIR/var_command.txt | $PARALLEL -j $THREADS

    # make vcf header
    i=`head -n 1 $WINDOWS | awk '{print $1":"$2".."$3}'`
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        grep \"^#\" $TEMP_DIR/$OUTBASE.$i.vcf > $TEMP_DIR/header.txt"
    fi
    grep "^#" $TEMP_DIR/$OUTBASE.$i.vcf > $TEMP_DIR/header.txt

    # merge the vcf region files
    if [[ "$ANNOTATE" -eq 1 ]]
    then
	FORK=""
	if [[ "$THREADS" -gt 1 ]]
	then
	    FORK="--fork $THREADS"
	fi

	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
            cat $TEMP_DIR/$OUTBASE.\"$chrom:\$start..\$end\".vcf | grep -v \"^#\" \\
                | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \\
                | $VEP \\
                $FORK \\
                -o STDOUT \\
                --force_overwrite \\
                --offline \\
                --no_stats \\
                --cache \\
                --dir_cache $VEP_CACHE_DIR \\
                --assembly $VEP_ASSEMBLY \\
                --species homo_sapiens \\
                --sift b \\
                --polyphen b \\
                --symbol \\
                --numbers \\
                --biotype \\
                --total_length \\
                --vcf \\
                --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \\
                | $BGZIP -c > $OUTPUT.vcf.gz"
	fi

	for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
	do
            # add "|| true" so it doesn't bail upon failure.
	    cat $TEMP_DIR/$OUTBASE."$i".vcf | grep -v "^#" || true
	done \
	    | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \
	    | $VEP \
            $FORK \
            -o STDOUT \
	    --force_overwrite \
            --offline \
            --no_stats \
            --cache \
            --dir_cache $VEP_CACHE_DIR \
            --assembly $VEP_ASSEMBLY \
            --species homo_sapiens \
            --sift b \
            --polyphen b \
            --symbol \
            --numbers \
            --biotype \
            --total_length \
            --vcf \
            --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \
	    | $BGZIP -c > $OUTPUT.vcf.gz

    else
	if [[ $VERBOSE -eq 1 ]]
        then
            echo -e "
            cat $TEMP_DIR/$OUTBASE.\"$chrom:\$start..\$end\".vcf | grep -v \"^#\" \\
                | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \\
                | $BGZIP -c > $OUTPUT.vcf.gz"
        fi

	for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
	do
            # if this fails then it bails out the script
	    cat $TEMP_DIR/$OUTBASE."$i".vcf | grep -v "^#" || true
	done \
	    | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \
	    | $BGZIP -c > $OUTPUT.vcf.gz
    fi

    # index the vcf
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
	$TABIX -f -p vcf $OUTPUT.vcf.gz"
    fi
    $TABIX -f -p vcf $OUTPUT.vcf.gz

    # clean up
    if [[ "$KEEP" -eq 0 ]]
    then
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        rm -r $TEMP_DIR"
	fi
	rm -r $TEMP_DIR
    fi

    echo "Done"

    # exit cleanly
    exit 0

}

function somatic() {
    function somatic_usage() {
        echo "
usage:   speedseq somatic [options] <reference.fa> <normal.bam> <tumor.bam>

positional args:
         reference.fa
                  genome reference fasta file
         normal.bam
                  germline BAM file(s) (comma separated BAMs from multiple libraries).
                  Must have readgroup information, and the SM readgroup tag will
                  be the VCF column header
         tumor.bam
                  tumor BAM file(s) (comma separated BAMs for multiple libraries).
                    Must have readgroup information, and the SM readgroup tag will
                    be the VCF column header

options:
         -o STR   output prefix [tumor.bam]
         -w FILE  BED file of windowed genomic intervals
         -t INT   threads [1]
         -F FLOAT Require at least this fraction of observations supporting
                    an alternate allele within a single individual in order
                    to evaluate the position [0.05]
         -C INT   Require at least this count of observations supporting
                    an alternate allele within a single individual in order
                    to evaluate the position [2]
         -S FLOAT minimum somatic score (SSC) for PASS [18]
         -q FLOAT minimum QUAL score to output non-passing somatic variants [1e-5]
         -T DIR   temp directory [./output_prefix.XXXXXXXXXXXX]

         -A       annotate the vcf with VEP
         -a       VEP assembly to use [GRCh37]
         -K FILE  path to speedseq.config file (default: same directory as speedseq)
         -v       verbose
         -k       keep tempory files
         -h       show this message
"
    }

    # Check options passed in.
    if test -z "$3"
    then
        somatic_usage
        exit 1
    fi

    # set defaults
    SPEEDSEQ_DIR=`dirname $0`
    CONFIG="$SPEEDSEQ_DIR/speedseq.config"
    REF="${@:(-3):1}"
    OUTPUT=`basename "$TUMOR_BAM"`
    THREADS=1
    MIN_ALT_FRACTION=0.05
    MIN_ALT_COUNT=2
    TEMP_DIR=""
    ANNOTATE=0
    MINQUAL=1e-5
    ONLY_SOMATIC=0
    SSC_THRES=18
    VERBOSE=1
    KEEP=0
    VEP_ASSEMBLY="GRCh37"

    while getopts ":ho:w:t:F:C:T:Aa:q:S:vkK:" OPTION
    do
        case "${OPTION}" in
            h)
                somatic_usage
                exit 1
                ;;
            o)
                OUTPUT="$OPTARG"
                ;;
            w)
                WINDOWS="$OPTARG"
                ;;
            t)
                THREADS="$OPTARG"
                ;;
            F)
                MIN_ALT_FRACTION="$OPTARG"
                ;;
	    C)
		MIN_ALT_COUNT="$OPTARG"
		;;
            T)
                TEMP_DIR="$OPTARG"
                ;;
            A)
                ANNOTATE=1
                ;;
	    a)
		VEP_ASSEMBLY="$OPTARG"
		;;
	    q)
		MINQUAL="$OPTARG"
		;;
	    S)
		SSC_THRES="$OPTARG"
		;;
            v)
                VERBOSE=1
                ;;
	    k)
		KEEP=1
		;;
	    K)
		CONFIG="$OPTARG"
		;;
        esac
    done

    NORMAL_BAM_STRING="${@:$((${OPTIND}+1)):1}"
    TUMOR_BAM_STRING="${@:$((${OPTIND}+2)):1}"
    TUMOR_BAM_LIST=(`echo $TUMOR_BAM_STRING | tr "," " "`)
    NORMAL_BAM_LIST=`echo $NORMAL_BAM_STRING | tr "," " "`

    if [[ -z $OUTPUT ]]
    then
	OUTPUT=`basename "${TUMOR_BAM_LIST[0]}"`
    fi
    OUTBASE=`basename "$OUTPUT"`
    OPTIND=0

    # Check the for the relevant binaries
    source_binaries $CONFIG

    if [[ ! -f "$FREEBAYES" ]]
    then
        somatic_usage
        echo -e "Error: freebayes executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$BGZIP" ]]
    then
        somatic_usage
        echo -e "Error: bgzip executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$TABIX" ]]
    then
        somatic_usage
        echo -e "Error: tabix executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$VEP" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
        somatic_usage
        echo -e "Error: VEP not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -d "$VEP_CACHE_DIR" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
        somatic_usage
        echo -e "Error: VEP cache directory not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$PARALLEL" ]]
    then
        somatic_usage
        echo -e "Error: parallel executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    fi

    # Check that the ref and bam files exist
    if [[ -z "$REF" ]] || [[ ! -f "$REF" ]]
    then
        somatic_usage
        echo -e "Error: Reference file $REF not found.\n"
        exit 1
    fi
    for TEST_BAM in ${NORMAL_BAM_LIST[@]} ${TUMOR_BAM_LIST[@]}
    do
        if [[ ! -f $TEST_BAM ]]
        then
            somatic_usage
            echo -e "Error: BAM file $TEST_BAM not found.\n"
            exit 1
        fi
    done

    echo "Calling somatic variants..."
    # make temporary directory
    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        create temporary directory"
    fi
    if [[ -z $TEMP_DIR ]]
    then
	TEMP_DIR=`mktemp -d ${OUTBASE}.XXXXXXXXXXXX`
    else
	mkdir -p $TEMP_DIR	
    fi

    # if no windows file, then make naive windows based on the chroms
    if [[ -z $WINDOWS ]]
    then
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        $SAMBAMBA view -H ${NORMAL_BAM_LIST[0]} | grep \"^@SQ\" | cut -f 2- | awk '{ gsub(\"^SN:\",\"\",\$1); gsub(\"^LN:\",\"\",\$2); print \$1\"\\\t0\\\t\"\$2; }' > $TEMP_DIR/windows.bed"
	fi
        $SAMBAMBA view -H ${NORMAL_BAM_LIST[0]} | grep "^@SQ" | cut -f 2- | awk '{ gsub("^SN:","",$1); gsub("^LN:","",$2); print $1"\t0\t"$2; }' > $TEMP_DIR/windows.bed
        WINDOWS="$TEMP_DIR/windows.bed"
    fi

    # write command to call variants on each of the windows in parallel
    NORMAL_BAM_FMT=`echo $NORMAL_BAM_STRING | tr "," " "`
    TUMOR_BAM_FMT=`echo $TUMOR_BAM_STRING | tr "," " "`

    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        $FREEBAYES -f $REF \\
            --pooled-discrete \\
            --min-repeat-entropy 1 \\
            --genotype-qualities \\
            --min-alternate-fraction ${MIN_ALT_FRACTION} \\
            --min-alternate-count ${MIN_ALT_COUNT} \\
            --region \$chrom:\$start..\$end \\
            $NORMAL_BAM_FMT $TUMOR_BAM_FMT \\
            | somatic_filter $MINQUAL $SSC_THRES $ONLY_SOMATIC \\
            > ${TEMP_DIR}/$OUTBASE.\$chrom:\$start..\$end.vcf"
    fi
    for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
    do
	# see the function somatic_filter for a more readable version of this awk command
	echo -e "$FREEBAYES -f $REF \
        --pooled-discrete \
        --genotype-qualities \
        --min-repeat-entropy 1 \
        --min-alternate-fraction ${MIN_ALT_FRACTION} \
        --min-alternate-count ${MIN_ALT_COUNT} \
        --region $i \
        $NORMAL_BAM_FMT $TUMOR_BAM_FMT \
        | awk -v ONLY_SOMATIC=\"$ONLY_SOMATIC\" -v MINQUAL=\"$MINQUAL\" -v SSC_THRES=\"$SSC_THRES\" 'BEGIN {NORMAL=10; TUMOR=11; GL_IDX=0;} { if (\$0~\"^#\") { print ; next; } if (! GL_IDX) { split(\$9,fmt,\":\") ; for (i=1;i<=length(fmt);++i) { if (fmt[i]==\"GL\") GL_IDX=i } } split(\$NORMAL,N,\":\"); split(N[GL_IDX],NGL,\",\"); split(\$TUMOR,T,\":\"); split(T[GL_IDX],TGL,\",\"); LOD_NORM=NGL[1]-NGL[2]; LOD_TUMOR_HET=TGL[2]-TGL[1]; LOD_TUMOR_HOM=TGL[3]-TGL[1]; if (LOD_TUMOR_HET > LOD_TUMOR_HOM) { LOD_TUMOR=LOD_TUMOR_HET } else { LOD_TUMOR=LOD_TUMOR_HOM } DQUAL=LOD_TUMOR+LOD_NORM; if (DQUAL>=SSC_THRES && \$NORMAL~\"^0/0\") { \$7=\"PASS\" ; \$8=\"SSC=\"DQUAL\";\"\$8 ; print } else if (!ONLY_SOMATIC && \$6>=MINQUAL && \$10~\"^0/0\" && ! match(\$11,\"^0/0\")) { \$8=\"SSC=\"DQUAL\";\"\$8 ; print } }' OFS=\"\t\" \
        > ${TEMP_DIR}/$OUTBASE.$i.vcf"
    done > $TEMP_DIR/var_command.txt

    # run the freebayes command in parallel
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        cat $TEMP_DIR/var_command.txt | $PARALLEL -j $THREADS"
    fi
    cat $TEMP_DIR/var_command.txt | $PARALLEL -j $THREADS

    # make vcf header
    i=`head -n 1 $WINDOWS | awk '{print $1":"$2".."$3}'`
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        grep \"^##\" $TEMP_DIR/$OUTBASE.$i.vcf \\
        | cat - <(echo '##INFO=<ID=SSC,Number=1,Type=Float,Description=\"Somatic score\">') <(grep \"^#CHROM\" $TEMP_DIR/$OUTBASE.$i.vcf) > $TEMP_DIR/header.txt"
    fi
    grep "^##" $TEMP_DIR/$OUTBASE.$i.vcf | cat - <(echo '##INFO=<ID=SSC,Number=1,Type=Float,Description="Somatic score">') <(grep "^#CHROM" $TEMP_DIR/$OUTBASE.$i.vcf) > $TEMP_DIR/header.txt

    # get the tumor and normal readgroups
    TUMOR_RG=`cat ${TEMP_DIR}/header.txt | tail -n 1 | cut -f 10`
    NORMAL_RG=`cat ${TEMP_DIR}/header.txt | tail -n 1 | cut -f 11`

    if [[ "$ANNOTATE" -eq 1 ]]
    then
        # merge the vcf region files, with VEP annotation
	FORK=""
	if [[ "$THREADS" -gt 1 ]]
	then
	    FORK="--fork $THREADS"
	fi

	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
            cat $TEMP_DIR/$OUTBASE.\"\$chrom:\$start..\$end\".vcf | grep -v \"^#\" \\
                | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \\
                | $VEP \\
                $FORK \\
                -o STDOUT \\
                --force_overwrite \\
                --offline \\
                --no_stats \\
                --cache \\
                --dir_cache $VEP_CACHE_DIR \\
                --assembly $VEP_ASSEMBLY \\
                --species homo_sapiens \\
                --sift b \\
                --polyphen b \\
                --symbol \\
                --numbers \\
                --biotype \\
                --total_length \\
                --vcf \\
                --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \\
                | $BGZIP -c > $OUTPUT.vcf.gz"
	fi    
	for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
	do
            # if this fails then it bails out the script, unless || true present
            cat $TEMP_DIR/$OUTBASE."$i".vcf | grep -v "^#" || true
	done \
            | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \
            | $VEP \
            $FORK \
            -o STDOUT \
	    --force_overwrite \
            --offline \
            --no_stats \
            --cache \
            --dir_cache $VEP_CACHE_DIR \
            --assembly $VEP_ASSEMBLY \
            --species homo_sapiens \
            --sift b \
            --polyphen b \
            --symbol \
            --numbers \
            --biotype \
            --total_length \
            --vcf \
            --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \
            | $BGZIP -c > $OUTPUT.vcf.gz
    else
        # merge the vcf region files, without VEP annotation
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        cat $TEMP_DIR/$OUTBASE.\"\$chrom:\$start..\$end\".vcf | grep -v \"^#\" \\
            | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \\
            | $BGZIP -c > $OUTPUT.vcf.gz"
	fi    
	for i in `cat $WINDOWS | awk '{print $1":"$2".."$3}'`
	do
            # if this fails then it bails out the script
            cat $TEMP_DIR/$OUTBASE."$i".vcf | grep -v "^#" || true
	done \
            | sort -k1,1 -k2,2n | cat $TEMP_DIR/header.txt - \
            | $BGZIP -c > $OUTPUT.vcf.gz
    fi

    # index the vcf
    if [[ $VERBOSE -eq 1 ]]
    then
	echo -e "
        $TABIX -f -p vcf $OUTPUT.vcf.gz"
    fi
    $TABIX -f -p vcf $OUTPUT.vcf.gz

    # produce PED file for GEMINI loading
    NORMAL_SAMPLE=`$SAMBAMBA view -H ${NORMAL_BAM_LIST[0]} | grep -m 1 "^@RG" | awk -v i=$i '{ for (j=1;j<=NF;++j) {if ($j~"^SM:") { gsub("^SM:","",$j); print $j } } }'`
    TUMOR_SAMPLE=`$SAMBAMBA view -H ${TUMOR_BAM_LIST[0]} | grep -m 1 "^@RG" | awk -v i=$i '{ for (j=1;j<=NF;++j) {if ($j~"^SM:") { gsub("^SM:","",$j); print $j } } }'`
    if [[ $VERBOSE -eq 1 ]]
    then
	echo "# Make PED file"
	echo "echo -e \"1\t$NORMAL_SAMPLE\tNone\tNone\t0\t1\n1\t$TUMOR_SAMPLE\tNone\tNone\t0\t2\" > $OUTPUT.ped"
    fi
    echo -e "1\t$NORMAL_SAMPLE\tNone\tNone\t0\t1\n1\t$TUMOR_SAMPLE\tNone\tNone\t0\t2" > $OUTPUT.ped

    # clean up
    if [[ "$KEEP" -eq 0 ]]
    then
	if [[ $VERBOSE -eq 1 ]]
	then
	    echo -e "
        rm -r $TEMP_DIR"
	fi
	rm -r $TEMP_DIR
    fi
	
    echo "Done"

    # exit cleanly
    exit 0
}

function sv() {
    function sv_usage() {
        echo "
usage:   speedseq sv [options]

sv options:
         -B FILE  full BAM file(s) (comma separated) (required)
         -S FILE  split reads BAM file(s) (comma separated) (required)
         -D FILE  discordant reads BAM files(s) (comma separated) (required)
         -R FILE  indexed reference genome fasta file (required)
         -o STR   output prefix [fullBam.bam]
         -t INT   threads [1] 
         -x FILE  BED file to exclude
         -g       genotype SV breakends with svtyper
         -d       calculate read-depth with CNVnator
         -w INT   CNVnator window size [100]
         -A       annotate the vcf with VEP
         -a       VEP assembly to use [GRCh37]
         -P       output LUMPY probability curves in VCF
         -m INT   minimum sample weight for a call [4]
         -r FLOAT trim threshold [0]
         -T DIR   temp directory [./output_prefix.XXXXXXXXXXXX]
         -k       keep temporary files

global options:
         -K FILE  path to speedseq.config file (default: same directory as speedseq)
         -v       verbose
         -h       show this message
"
    }

    # set defaults
    SPEEDSEQ_DIR=`dirname $0`
    CONFIG="$SPEEDSEQ_DIR/speedseq.config"
    THREADS=1
    ANNOTATE=0
    VEP_ASSEMBLY="GRCh37"
    MIN_SAMPLE_WEIGHT=4
    TRIM_THRES=0
    EXCLUDE_BED=
    TEMP_DIR=""
    GENOTYPE=0
    READDEPTH=0
    WINDOW_SIZE=100
    PROBCURVE=0
    VERBOSE=1
    KEEP=0
    OUTPUT=""
    MAX_SPLIT_COUNT=2
    MIN_NON_OVERLAP=20

    while getopts ":hB:S:D:R:o:m:r:x:T:t:Aa:dw:gPkvK:" OPTION
    do
	case "${OPTION}" in
            h)
                sv_usage
                exit 1
                ;;
	    B)
		FULL_BAM_STRING="$OPTARG"
		;;
	    S)
		SPL_BAM_STRING="$OPTARG"
		SPL_BAM_STRING_FMT="-S $OPTARG"
		;;
	    D)
		DISC_BAM_STRING="$OPTARG"
		DISC_BAM_STRING_FMT="-D $OPTARG"
		;;
	    R)
		REF="$OPTARG"
		;;
            o)
                OUTPUT="$OPTARG"
                ;;
            m)
                MIN_SAMPLE_WEIGHT="$OPTARG"
                ;;
            r)
                TRIM_THRES="$OPTARG"
                ;;
            x)
	        EXCLUDE_BED="$OPTARG"
		EXCLUDE_BED_FMT="-x $EXCLUDE_BED"
		;;  
            T)
                TEMP_DIR="$OPTARG"
                ;;
	    t)
		THREADS="$OPTARG"
		;;
	    A)
		ANNOTATE=1
		;;
	    a)
		VEP_ASSEMBLY="$OPTARG"
		;;
	    d)
		READDEPTH=1
		;;
	    w)
		WINDOW_SIZE="$OPTARG"
		;;
	    g)
		GENOTYPE=1
		;;
	    P)
		PROBCURVE=1
		PROBCURVE_FMT="-P"
		;;
            v)
                VERBOSE=1
		;;
	    k)
		KEEP=1
		;;
	    K)
		CONFIG="$OPTARG"
		;;
	esac
    done

    # parse the BAM strings
    FULL_BAM_LIST=(`echo $FULL_BAM_STRING | tr "," " "`)
    SPL_BAM_LIST=(`echo $SPL_BAM_STRING | tr "," " "`)
    DISC_BAM_LIST=(`echo $DISC_BAM_STRING | tr "," " "`)

    OPTIND=0

    # Check the for the relevant binaries
    source_binaries $CONFIG

    if [[ ! -f "$LUMPY" ]]
    then
	sv_usage
        echo -e "Error: lumpy executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$LUMPYEXPRESS" ]]
    then
	sv_usage
        echo -e "Error: lumpyexpress executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f  "$PAIREND_DISTRO" ]]
    then
	sv_usage
        echo -e "Error: pairend_distro.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f  "$SAMBAMBA" ]]
    then
	sv_usage
        echo -e "Error: sambamba executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$VEP" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
	sv_usage
        echo -e "Error: VEP not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -d "$VEP_CACHE_DIR" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
	sv_usage
        echo -e "Error: VEP cache directory not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$VAWK" ]] && [[ "$ANNOTATE" -eq 1 ]]
    then
	sv_usage
        echo -e "Error: vawk executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$SAMBLASTER" ]] && [[ -z "${DISC_BAM_STRING}${SPL_BAM_STRING}" ]]
    then
	sv_usage
        echo -e "Error: samblaster executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$BAMFILTERRG" ]]
    then
	sv_usage
	echo -e "Error: bamfilterrg.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$BAMLIBS" ]]
    then
	sv_usage
	echo -e "Error: bamlibs.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    fi

    # if genotyping requested, look for svtyper
    if [[ "$GENOTYPE" -eq 1 ]] && [[ ! -f "$SVTYPER" ]]
    then
	sv_usage
        echo -e "Error: svtyper executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    fi

    # if CNV read-depth requested, look for cnvnator executables
    if [[ "$READDEPTH" -eq 1 ]]
    then
	if [[ ! -f "$CNVNATOR" ]]
	then
	    sv_usage
            echo -e "Error: cnvnator executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
            exit 1
	elif [[ ! -f "$CNVNATOR_WRAPPER" ]]
	then
	    sv_usage
            echo -e "Error: cnvnator_wrapper.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
            exit 1
	elif [[ ! -f "$ANNOTATE_RD" ]]
	then
	    sv_usage
            echo -e "Error: annotate_rd.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
            exit 1
	fi
    fi

    # check for required python modules (pysam, numpy, scipy, etc)
    check_python_modules $PYTHON

    # Check that the required files exist
    if [[ ! -f $REF ]]
    then
	sv_usage
	echo -e "Error: reference fasta file $REF not found\n"
	exit 1
    fi
    if [[ ! -f $REF.fai && ! -f $(echo ${REF%*.*}).fai ]]
    then
	sv_usage
	echo -e "Error: reference fasta file $REF not indexed. Please run samtools faidx on the .fasta file\n"
	exit 1
    fi
    if [[ ${#FULL_BAM_LIST[@]} -eq 0 ]]
    then
	sv_usage
	echo -e "Error: -B is required\n"
	exit 1
    fi

    for TEST_BAM in ${FULL_BAM_LIST[@]} ${SPL_BAM_LIST[@]} ${DISC_BAM_LIST[@]}
    do
        if [[ ! -f $TEST_BAM ]]
        then
            sv_usage
            echo -e "Error: BAM file $TEST_BAM not found.\n"
            exit 1
        fi
    done

    # default OUTPUT if not provided
    if test -z "$OUTPUT"
    then
	OUTPUT=`basename "${FULL_BAM_LIST[0]}"`
    fi
    OUTBASE=`basename "$OUTPUT"`

    # make temporary directory
    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        create temporary directory"
    fi
    if [[ -z $TEMP_DIR ]]
    then
	TEMP_DIR=`mktemp -d ${OUTBASE}.XXXXXXXXXXXX`
    else
	mkdir -p $TEMP_DIR	
    fi

    # run lumpy express
    echo -e "\nRunning LUMPY express"
    $LUMPYEXPRESS \
	-B $FULL_BAM_STRING \
	$SPL_BAM_STRING_FMT \
	$DISC_BAM_STRING_FMT \
	-o $TEMP_DIR/$OUTBASE.sv.vcf \
	$EXCLUDE_BED_FMT \
	$PROBCURVE_FMT \
	-r $TRIM_THRES \
	-m $MIN_SAMPLE_WEIGHT \
	-T ${TEMP_DIR}/temp_lumpyexpress \
	-K $CONFIG \
	-v \
	-k

    # genotype with SVTyper
    if [[ GENOTYPE -eq 1 ]]
    then
	for i in $( seq 0 $(( ${#FULL_BAM_LIST[@]}-1 )) )
	do
	    FULL_BAM=${FULL_BAM_LIST[$i]}
	    SPL_BAM=${SPL_BAM_LIST[$i]}
	    FULL_BASE=`basename "$FULL_BAM"`

	    if [[ "$VERBOSE" -eq 1 ]]
	    then
		echo "# genotype structural variants"
		echo -e "$PYTHON $SVTYPER -q -i $TEMP_DIR/$OUTBASE.sv.vcf -B $FULL_BAM -S $SPL_BAM > $TEMP_DIR/$OUTBASE.sv.gt.vcf ; mv $TEMP_DIR/$OUTBASE.sv.gt.vcf $TEMP_DIR/$OUTBASE.sv.vcf"
	    fi

	    # genotype structural variants
	    $PYTHON $SVTYPER -q -i $TEMP_DIR/$OUTBASE.sv.vcf -B $FULL_BAM -S $SPL_BAM > $TEMP_DIR/$OUTBASE.sv.gt.vcf
	    mv $TEMP_DIR/$OUTBASE.sv.gt.vcf $TEMP_DIR/$OUTBASE.sv.vcf
	done
    fi

    # run cnvnator
    if [[ "$READDEPTH" -eq 1 ]]
    then
	echo "Calculating read depth"

	for i in $( seq 0 $(( ${#FULL_BAM_LIST[@]}-1 )) )
	do
	    FULL_BAM=${FULL_BAM_LIST[$i]}
	    check_valid_cram_or_bam ${FULL_BAM}
	    FULL_BASE=`basename $FULL_BAM`
	    CRAM_OPTS=""
	    if is_cram $FULL_BAM; then
	        CRAM_OPTS="-C"
	    fi
	    SAMPLE=`$SAMBAMBA view $CRAM_OPTS -H $FULL_BAM | grep -m 1 "^@RG" | awk -v i=$i '{ for (j=1;j<=NF;++j) {if ($j~"^SM:") { gsub("^SM:","",$j); print $j } } }'`

	    if [[ "$VERBOSE" -eq 1 ]]
	    then
		echo "
    # run cnvnator
    $PYTHON $CNVNATOR_WRAPPER --samtools $SAMTOOLS --cnvnator $CNVNATOR -T $TEMP_DIR/cnvnator-temp -t $THREADS -w $WINDOW_SIZE -b ${FULL_BAM} -o $TEMP_DIR/$FULL_BASE.readdepth -c $CNVNATOR_CHROMS_DIR
		"
	    fi
	    $PYTHON $CNVNATOR_WRAPPER --samtools $SAMTOOLS --cnvnator $CNVNATOR -T $TEMP_DIR/cnvnator-temp -t $THREADS -w $WINDOW_SIZE -b ${FULL_BAM} -o $TEMP_DIR/$FULL_BASE.readdepth -c $CNVNATOR_CHROMS_DIR

	    # Calculate read-depth of LUMPY calls
	    if [[ "$VERBOSE" -eq 1 ]]
	    then
		echo "
    # Calculate read-depth of LUMPY calls
    $PYTHON $ANNOTATE_RD --cnvnator $CNVNATOR -s $SAMPLE -w $WINDOW_SIZE -r $TEMP_DIR/cnvnator-temp/${FULL_BASE}.hist.root -v $TEMP_DIR/$OUTBASE.sv.vcf > $TEMP_DIR/$OUTBASE.sv.rd.vcf
		"
	    fi
	    $PYTHON $ANNOTATE_RD --cnvnator $CNVNATOR -s $SAMPLE -w $WINDOW_SIZE -r $TEMP_DIR/cnvnator-temp/${FULL_BASE}.hist.root -v $TEMP_DIR/$OUTBASE.sv.vcf > $TEMP_DIR/$OUTBASE.rd.sv.vcf

	    if [[ "$VERBOSE" -eq 1 ]]
	    then
		echo "mv $TEMP_DIR/$OUTBASE.rd.sv.vcf $TEMP_DIR/$OUTBASE.rd.sv.vcf"
		echo "mv $TEMP_DIR/$FULL_BASE.readdepth.txt $OUTPUT.$FULL_BASE.sv.readdepth.txt"
		echo "mv $TEMP_DIR/$FULL_BASE.readdepth.bed $OUTPUT.$FULL_BASE.sv.readdepth.bed"
	    fi
	    mv $TEMP_DIR/$OUTBASE.rd.sv.vcf $TEMP_DIR/$OUTBASE.sv.vcf
	    mv $TEMP_DIR/$FULL_BASE.readdepth.txt $OUTPUT.sv.$FULL_BASE.readdepth.txt
	    mv $TEMP_DIR/$FULL_BASE.readdepth.bed $OUTPUT.sv.$FULL_BASE.readdepth.bed
	done
    fi

    # Annotate structural variants with VEP
    if [[ "$ANNOTATE" -eq 1 ]]
    then
	FORK=""
	if [[ "$THREADS" -gt 1 ]]
	then
	    FORK="--fork $THREADS"
	fi

	if [[ "$VERBOSE" -eq 1 ]]
	then
	    echo -e "
# Annotate structural variants with VEP
cat  $TEMP_DIR/$OUTBASE.sv.vcf \\
    | $PYTHON $VAWK --header '(\$1<=22 || \$1==\"X\" || \$1==\"Y\") && (I\$SVTYPE==\"BND\" || (I\$SVLEN<=50000 && I\$SVLEN>=-50000))' \\
    | $VEP \\
    $FORK \\
    -o STDOUT \\
    --force_overwrite \\
    --format vcf \\
    --offline \\
    --no_stats \\
    --cache \\
    --dir_cache $VEP_CACHE_DIR \\
    --assembly $VEP_ASSEMBLY \\
    --species homo_sapiens \\
    --sift b \\
    --polyphen b \\
    --symbol \\
    --numbers \\
    --biotype \\
    --total_length \\
    --vcf \\
    --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \\
    | cat - <(cat $TEMP_DIR/$OUTBASE.sv.vcf | $PYTHON $VAWK '(\$1>22 && \$1!=\"X\" && \$1!=\"Y\") || (I\$SVTYPE!=\"BND\" && (I\$SVLEN>50000 || I\$SVLEN<-50000))') \\
    > $TEMP_DIR/$OUTBASE.vep.sv.vcf
    "
	fi

	cat  $TEMP_DIR/$OUTBASE.sv.vcf \
	    | $PYTHON $VAWK --header '($1<=22 || $1=="X" || $1=="Y") && (I$SVTYPE=="BND" || (I$SVLEN<=50000 && I$SVLEN>=-50000))' \
	    | $VEP \
            $FORK \
            -o STDOUT \
            --force_overwrite \
	    --format vcf \
            --offline \
            --no_stats \
            --cache \
            --dir_cache $VEP_CACHE_DIR \
	    --assembly $VEP_ASSEMBLY \
            --species homo_sapiens \
            --sift b \
            --polyphen b \
            --symbol \
            --numbers \
            --biotype \
            --total_length \
            --vcf \
            --fields Consequence,Codons,Amino_acids,Gene,SYMBOL,Feature,EXON,PolyPhen,SIFT,Protein_position,BIOTYPE \
	    | cat - <(cat $TEMP_DIR/$OUTBASE.sv.vcf | $PYTHON $VAWK '($1>22 && $1!="X" && $1!="Y") || (I$SVTYPE!="BND" && (I$SVLEN>50000 || I$SVLEN<-50000))') \
	    > $TEMP_DIR/$OUTBASE.vep.sv.vcf

	mv $TEMP_DIR/$OUTBASE.vep.sv.vcf $TEMP_DIR/$OUTBASE.sv.vcf
    fi

    # write output vcf file
    cat $TEMP_DIR/$OUTBASE.sv.vcf | awk '{ print; if ($0~"^#CHROM") exit; }' > $TEMP_DIR/header.txt
    cat $TEMP_DIR/$OUTBASE.sv.vcf \
	| grep -v "^#" \
	| sort -k1,1 -k2,2n \
	| cat $TEMP_DIR/header.txt - \
	| $BGZIP -c \
	> $OUTPUT.sv.vcf.gz
    $TABIX -p vcf $OUTPUT.sv.vcf.gz

    # clean up
    if [[ "$KEEP" -eq 0 ]]
    then
	rm -r ${TEMP_DIR}
    fi

    echo "SV calling done."
    
    # exit cleanly
    exit 0
}

function realign() {
    function realign_usage() {
	echo "
usage:   speedseq realign [options] <reference.fa> <in1.bam> [in2.bam [...]]

positional args:
         reference.fa
                  fasta file (indexed with bwa)
         in.bam   BAM file(s) (must contain read group tags)

alignment options:
         -o STR   output prefix [in.realign]
         -I FLOAT[,FLOAT[,INT[,INT]]]
                  specify the mean, standard deviation (10% of the mean if absent), max
                    (4 sigma from the mean if absent) and min of the insert size distribution.
                    FR orientation only. [inferred]
         -n       rename reads for smaller file size
         -t INT   threads [1]
         -T DIR   temp directory [./output_prefix.XXXXXXXXXXXX]
         -R STR   read group header line such as \"@RG\tID:id\tSM:samplename\tLB:lib\"
                    WARNING: By default SpeedSeq will automatically parse readgroup info
                    from the input BAM headers. This option will supercede that info and
                    should therefore only be used when input BAM(s) lack readgroups or when
                    custom editing of readgroups is desired.

samblaster options:
         -i       include duplicates in splitters and discordants
         -c INT   maximum number of split alignments for a read to be included in splitter file [2]
         -m INT   minimum non-overlapping base pairs between two alignments for a read to be included in splitter file [20]

sambamba options:
         -M       amount of memory in GB to be used for sorting [20]

global options:
         -K FILE  path to speedseq.config file (default: same directory as speedseq)
         -v       verbose
         -h       show this message
"
    }

    # Check options passed in.
    if test -z "$2"
    then
	realign_usage
	exit 1
    fi

    # set defaults
    SPEEDSEQ_DIR=`dirname $0`
    CONFIG="$SPEEDSEQ_DIR/speedseq.config"
    INTERLEAVED=0
    OUTPUT=""
    INCLUDE_DUPS="--excludeDups"
    MAX_SPLIT_COUNT=2
    MIN_NON_OVERLAP=20
    THREADS=1
    TEMP_DIR=""
    VERBOSE=1
    INS_DIST=""
    RENAME=""
    SORT_MEM=20 # amount of memory for sorting, in gigabytes
    RG_FMT=""

    while getopts ":hw:o:pic:m:M:t:T:I:nvK:R:" OPTION
    do
	case "${OPTION}" in
	    h)
		realign_usage
		exit 1
		;;
	    p)
		INTERLEAVED=1
		;;
	    o)
		OUTPUT="$OPTARG"
		;;
	    i)
		INCLUDE_DUPS=""
		;;
	    c)
		MAX_SPLIT_COUNT="$OPTARG"
		;;
	    m)
		MIN_NON_OVERLAP="$OPTARG"
		;;
	    M)
		SORT_MEM="$OPTARG"
		;;
	    n)
		RENAME="-n"
		;;
	    t)
		THREADS="$OPTARG"
		;;
	    T)
		TEMP_DIR="$OPTARG"
		;;
	    I)
		INS_DIST="-I $OPTARG"
		;;
	    v)
		VERBOSE=1
		;;
	    K)
		CONFIG="$OPTARG"
		;;
            R)
                RG="$OPTARG"
                RG_FMT="-R '$OPTARG'"
                ;;
	esac
    done

    REF="${@:${OPTIND}:1}"
    BAM_STRING="${@:$((${OPTIND}+1))}"
    BAM_LIST=($BAM_STRING)

    if [[ -z "$OUTPUT" ]]
    then
	OUTPUT=`basename "${BAM_LIST[0]}" ".bam"`".realign"
    fi

    # Check that the ref files exists
    if [[ -z "$REF" ]] || [[ ! -f "$REF" ]]
    then
	realign_usage
	echo -e "Error: Reference file $REF not found.\n"
	exit 1
    fi

    # Check that the input BAMs exist
    for TEST_BAM in ${BAM_LIST[@]}
    do
        if [[ ! -f $TEST_BAM ]]
        then
            realign_usage
            echo -e "Error: BAM file $TEST_BAM not found.\n"
            exit 1
        fi
    done

    # Check that SORT_MEM > 2
    if [[ "$SORT_MEM" -lt 3 ]]
    then
	realign_usage
        echo -e "Error: -M must be greater than 2"
	exit 1
    fi

    # Check the for the relevant binaries
    source_binaries $CONFIG
    if [[ ! -f "$BWA" ]]
    then
	realign_usage
        echo -e "Error: bwa executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f  "$SAMBLASTER" ]]
    then
	realign_usage
        echo -e "Error: samblaster executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$SAMBAMBA" ]]
    then
	realign_usage
        echo -e "Error: sambamba executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    elif [[ ! -f "$MBUFFER" ]]
    then
	realign_usage
	echo -e "Error: mbuffer executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$BAMTOFASTQ" ]]
    then
	realign_usage
	echo -e "Error: bamtofastq.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$BAMHEADRG" ]]
    then
	realign_usage
	echo -e "Error: bamheadrg.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$BAMCLEANHEADER" ]]
    then
	realign_usage
	echo -e "Error: bamcleanheader.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$BAMLIBS" ]]
    then
	realign_usage
	echo -e "Error: bamlibs.py executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
	exit 1
    elif [[ ! -f "$PARALLEL" ]]
    then
        realign_usage
	echo parallel is $PARALLEL
        echo -e "Error: parallel executable not found. Please set path in $SPEEDSEQ_DIR/speedseq.config file\n"
        exit 1
    fi

    # check for required python modules (pysam, numpy, scipy, etc)
    check_python_modules $PYTHON

    # set the output name
    OUTBASE=`basename "$OUTPUT"`
    # create temp directory and pipes
    if [[ -z $TEMP_DIR ]]
    then
	TEMP_DIR=`mktemp -d ${OUTBASE}.XXXXXXXXXXXX`
    fi

    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        mkdir -p $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
        mkfifo $TEMP_DIR/spl_pipe $TEMP_DIR/disc_pipe"
    fi

    if [[ $VERBOSE -eq 1 ]]
    then
	echo "
        mkdir -p $TEMP_DIR $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
        mkdir -p $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
        mkfifo $TEMP_DIR/spl_pipe $TEMP_DIR/disc_pipe $TEMP_DIR/fq_pipe"
    fi
    mkdir -p $TEMP_DIR $TEMP_DIR/full $TEMP_DIR/spl $TEMP_DIR/disc
    if [[ ! -e $TEMP_DIR/spl_pipe ]]
    then
	mkfifo $TEMP_DIR/spl_pipe
    fi
    if [[ ! -e $TEMP_DIR/disc_pipe ]]
    then
	mkfifo $TEMP_DIR/disc_pipe
    fi
    if [[ ! -e $TEMP_DIR/fq_pipe ]]
    then
	mkfifo $TEMP_DIR/fq_pipe
    fi
    FQ="$TEMP_DIR/fq_pipe"

    # make a sloppy "merged" header
    if [[ -z "$RG_FMT" ]]
    then
	$PYTHON $BAMCLEANHEADER $BAM_STRING > $TEMP_DIR/header.txt
    else
	> $TEMP_DIR/header.txt
    fi

    # SHOULD ADD A CHECK HERE TO MAKE SURE EACH INDIVIDUAL BAM CONTAINS READGROUPS

    # parse the libraries in the BAM header to extract readgroups from the same library
    if [[ -z $REALIGN_RG_LIST ]]
    then
	REALIGN_RG_LIST+=(`$PYTHON $BAMLIBS -S $TEMP_DIR/header.txt`)	
    fi

    # if input BAM lacks RG info, generate it from output filename
    if [[ ${#REALIGN_RG_LIST[@]} -eq 0 ]]
    then
	# realign_usage
        # exit 1
        echo -e "Warning: BAM headers lack read group fields (@RG)\n"
	REALIGN_RG_LIST=""
	# use the supplied readgroup from the command line if available, otherwise
	# create readgroup from the output name
	if [[ -z "$RG_FMT" ]]
	then
	    RG_FMT="-R '@RG\tID:$OUTPUT\tSM:$OUTPUT\tLB:$OUTPUT'"
	fi
    fi
    
    # if user supplies command line readgroup info, then override existing RG info in input BAMs
    if [[ -z "$RG_FMT" ]]
    then
	RETAIN_INPUT_RG="-C"
    else
	RETAIN_INPUT_RG=""
    fi

    # Check for BWA index of the reference
    if [[ ! -f "$REF.bwt" ]] || [[ ! -f "$REF.pac" ]] || [[ ! -f "$REF.ann" ]] || [[ ! -f "$REF.amb" ]] || [[ ! -f "$REF.sa" ]]
    then
	echo "Warning: Reference file not indexed with BWA. Indexing now..."
	$BWA index $REF
	echo "Done"
    fi

    echo "Aligning..."
    for i in $( seq 0 $(( ${#REALIGN_RG_LIST[@]}-1 )) )
    do
	REALIGN_RG=${REALIGN_RG_LIST[i]}
	if [[ -z "$REALIGN_RG" ]]
	then
	    REALIGN_RG_FMT=""
	else
	    REALIGN_RG_FMT="-r ${REALIGN_RG_LIST[i]}"
	fi
	echo "             Realigning readgroups: $REALIGN_RG"

	# alignment command
	if [[ $VERBOSE -eq 1 ]]
	    then
	        echo -e "
            $PYTHON $BAMTOFASTQ $REALIGN_RG_FMT $RENAME $BAM_STRING | $MBUFFER -q -m 1G > $FQ
            $BWA mem -t $THREADS $RETAIN_INPUT_RG -p $INS_DIST $RG_FMT $REF $FQ \\
                | $PYTHON $BAMHEADRG -d $TEMP_DIR/header.txt $REALIGN_RG_FMT \\
                | $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe \\
                | $SAMBAMBA view -S -f bam -l 0 /dev/stdin \\
                | $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $TEMP_DIR/$OUTBASE.$(($i+1)).bam /dev/stdin
	    gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe \\
		| $SAMBAMBA view -S -f bam -l 0 /dev/stdin \\
                | $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $TEMP_DIR/$OUTBASE.$(($i+1)).splitters.bam /dev/stdin
	    gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe \\
		| $SAMBAMBA view -S -f bam -l 0 /dev/stdin \\
                | $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $TEMP_DIR/$OUTBASE.$(($i+1)).discordants.bam /dev/stdin"
	fi

	echo -e "
            $PYTHON $BAMTOFASTQ $REALIGN_RG_FMT $RENAME $BAM_STRING | $MBUFFER -q -m 1G > $FQ
            $BWA mem -t $THREADS $RETAIN_INPUT_RG -p $INS_DIST $RG_FMT $REF $FQ \
                | $PYTHON $BAMHEADRG -d $TEMP_DIR/header.txt $REALIGN_RG_FMT \
                | $SAMBLASTER $INCLUDE_DUPS --addMateTags --maxSplitCount $MAX_SPLIT_COUNT --minNonOverlap $MIN_NON_OVERLAP --splitterFile $TEMP_DIR/spl_pipe --discordantFile $TEMP_DIR/disc_pipe \
                | $SAMBAMBA view -S -f bam -l 0 /dev/stdin \
                | $SAMBAMBA sort -t $THREADS -m $((${SORT_MEM}-2))G --tmpdir=$TEMP_DIR/full -o $TEMP_DIR/$OUTBASE.$(($i+1)).bam /dev/stdin
	    gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/spl_pipe \
		| $SAMBAMBA view -S -f bam -l 0 /dev/stdin \
                | $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/spl -o $TEMP_DIR/$OUTBASE.$(($i+1)).splitters.bam /dev/stdin
	    gawk '{ if (\$0~\"^@\") { print; next } else { \$10=\"*\"; \$11=\"*\"; print } }' OFS=\"\\t\" $TEMP_DIR/disc_pipe \
		| $SAMBAMBA view -S -f bam -l 0 /dev/stdin \
                | $SAMBAMBA sort -t 4 -m 1G --tmpdir=$TEMP_DIR/disc -o $TEMP_DIR/$OUTBASE.$(($i+1)).discordants.bam /dev/stdin
            " | $PARALLEL -j 4
    done

    # if only 1 library, then rename files and index
    if [[ ${#REALIGN_RG_LIST[@]} -eq 1 ]]
    then
	mv $TEMP_DIR/$OUTBASE.1.bam $OUTPUT.bam
	mv $TEMP_DIR/$OUTBASE.1.discordants.bam $OUTPUT.discordants.bam
	mv $TEMP_DIR/$OUTBASE.1.splitters.bam $OUTPUT.splitters.bam

	echo "
	$SAMBAMBA index $OUTPUT.bam
	$SAMBAMBA index $OUTPUT.discordants.bam
	$SAMBAMBA index $OUTPUT.splitters.bam
	" | $PARALLEL -j 3

    else
	MERGE_FULL=""
	MERGE_DISCORDANTS=""
	MERGE_SPLITTERS=""
	for i in $( seq 0 $(( ${#REALIGN_RG_LIST[@]}-1 )) )
	do
	    MERGE_FULL="$MERGE_FULL $TEMP_DIR/$OUTBASE.$(($i+1)).bam"
	    MERGE_DISCORDANTS="$MERGE_DISCORDANTS $TEMP_DIR/$OUTBASE.$(($i+1)).discordants.bam"
	    MERGE_SPLITTERS="$MERGE_SPLITTERS $TEMP_DIR/$OUTBASE.$(($i+1)).splitters.bam"
	done
	
	if [[ $VERBOSE -eq 1 ]]
	then
            echo "
            $SAMBAMBA merge -t $THREADS $OUTPUT.bam $MERGE_FULL
            $SAMBAMBA merge -t $THREADS $OUTPUT.discordants.bam $MERGE_DISCORDANTS
            $SAMBAMBA merge -t $THREADS $OUTPUT.splitters.bam $MERGE_SPLITTERS
            rm $MERGE_FULL $MERGE_DISCORDANTS $MERGE_SPLITTERS"
	fi
	$SAMBAMBA merge -t $THREADS $OUTPUT.bam $MERGE_FULL
	$SAMBAMBA merge -t $THREADS $OUTPUT.discordants.bam $MERGE_DISCORDANTS
	$SAMBAMBA merge -t $THREADS $OUTPUT.splitters.bam $MERGE_SPLITTERS
	rm $MERGE_FULL $MERGE_DISCORDANTS $MERGE_SPLITTERS
	
        # index the files
	if [[ $VERBOSE -eq 1 ]]
	then
            echo -e "
            $SAMBAMBA index $OUTPUT.bam
            $SAMBAMBA index $OUTPUT.discordants.bam
            $SAMBAMBA index $OUTPUT.splitters.bam"
	fi
	echo "
        $SAMBAMBA index $OUTPUT.bam
        $SAMBAMBA index $OUTPUT.discordants.bam
        $SAMBAMBA index $OUTPUT.splitters.bam
        " | $PARALLEL -j 3
    fi

    # clean up
    rm -r $TEMP_DIR

    echo "Done"

    # exit cleanly
    exit 0
}

# Show usage when there are no arguments.
if test -z "$1"
then
    usage
    exit 1
fi

while getopts "K:h" OPTION
do
    case $OPTION in
        h)
            usage
            exit 1
            ;;
	K)
	    ;;
        ?)
            usage
            exit
            ;;
    esac
done

# call the function
case "$1" in 
    'align')
	align "${@:2}"
	;;
    'var')
	var "${@:2}"
	;;
    'somatic')
	somatic "${@:2}"
	;;
    'sv')
	sv "${@:2}"
	;;
    'realign')
	realign "${@:2}"
	;;
    *)
	usage
	echo -e "Error: command \"$1\" not recognized\n"
	exit 1
esac

## END SCRIPT

#                             .       .
#                            / `.   .' \
#                    .---.  <    > <    >  .---.
#                    |    \  \ - ~ ~ - /  /    |
#                     ~-..-~             ~-..-~
#                 \~~~\.'                    `./~~~/
#       .-~~^-.    \__/                        \__/
#     .'  O    \     /               /       \  \
#    (_____,    `._.'               |         }  \/~~~/
#     `----.          /       }     |        /    \__/
#           `-.      |       /      |       /      `. ,~~|
#               ~-.__|      /_ - ~ ^|      /- _      `..-'   f: f:
#                    |     /        |     /     ~-.     `-. _||_||_
#                    |_____|        |_____|         ~ - . _ _ _ _ _>
#
