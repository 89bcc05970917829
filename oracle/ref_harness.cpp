/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A line-protocol driver around the REAL reference objects (compiled in place from
 * /root/reference/src by oracle/Makefile into oracle/_ref/).  It contains no algorithm of its
 * own: every answer it prints comes out of reference code.  Linked reference files:
 * editdistance, filter, polyx, stats, filterresult, read, options, fastareader, jsonreporter,
 * htmlreporter, writer, threadconfig (+ the system's libdeflate.so.0, which Writer calls for .gz names).  NOT linked (they include Google Highway / ISA-L headers the image lacks):
 * adaptertrimmer, sequence, fastqreader, seprocessor, evaluator, main.
 *
 * Protocol (stdin, one command per line, fields separated by one space, strings carry a
 * leading '=' so that the empty string is "="):
 *   ED =a =b                                   -> edit_distance
 *   TAC front tail cf ct wf qf wt qt =seq =qual -> Filter::trimAndCut
 *   PX minlen =seq =qual                        -> PolyX::trimPolyX
 *   PF qf qq up nbl npl aq lf rl ml cf cp =seq =qual -> Filter::passFilter
 *   TF n =seq =qual / RS n =seq =qual           -> Read::trimFront / Read::resize
 *   BG start len =name =seq =strand =qual       -> Read::breakByGap + appendToString
 *   TAG code =name =seq =strand =qual           -> Read::appendToStringWithTag(FAILED_TYPES[code])
 *   LQR window quality =seq =qual               -> Filter::detectLowQualityRegions
 *   BRK be bw bq me mw mq =name =seq =strand =qual -> the --break / --mask stage (detect, breakByRegions,
 *                                                  maskRegionWithN) + appendToString of what comes out
 *   JSON block: J_BEGIN threads seqlen isrna adapter_enabled polyx complexity =start =end
 *               J_PRE =seq =qual | J_POST =seq =qual | J_FR code | J_AD =key | J_ART bases
 *               | J_PXT base len | J_END =path   -> Stats/FilterResult/JsonReporter::report
 *               | J_ENDH lf maxlen =json =html =title words...  -> the same plus calcLengthHistogram and
 *                 HtmlReporter::report (lf / maxlen = Options::lengthFilter.enabled / .maxLength)
 *   FA =path                                    -> FastaReader::readAll: the contig count, then "=hex(header) =hex(sequence)"
 *                                                  per contig in map order
 *   --split / --split_by_lines block (real ThreadConfig + Writer objects, one per worker, as
 *   SingleEndProcessor::initConfig / processSingleEnd / ~ThreadConfig drive them, src/seprocessor.cpp:54-63,297-316):
 *               S_BEGIN threads bylines number size digits compression =out -> ThreadConfig(opt, t) x threads,
 *                       initWriterForSplit() each
 *               S_PACK worker n =text -> getWriter1()->writeString(text) (when --out is set), markProcessed(n);
 *                       prints canBeStopped() afterwards
 *               S_END   -> the ThreadConfig destructors (writeEmptyFilesForSplitting, writers flushed and closed)
 */
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "editdistance.h"
#include "fastareader.h"
#include "filter.h"
#include "filterresult.h"
#include "htmlreporter.h"
#include "jsonreporter.h"
#include "options.h"
#include "polyx.h"
#include "read.h"
#include "stats.h"
#include "threadconfig.h"

using namespace std;

/* the two globals main.cpp normally owns (the reference's test/globals.cpp does the same) */
string command;
mutex logmtx;

static vector<string> split(const string& s) {
    vector<string> out;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(' ', i);
        if (j == string::npos) j = s.size();
        out.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return out;
}
static string hex_of(const string& s) { /* (headers and sequences may hold any byte) */
    static const char* d = "0123456789abcdef";
    string o;
    for (unsigned char c : s) {
        o += d[c >> 4];
        o += d[c & 15];
    }
    return o;
}
static string str(const string& tok) { return tok.empty() ? string() : tok.substr(1); }

struct JsonJob {
    Options opt;
    vector<Stats*> pre, post;
    vector<FilterResult*> fr;
    int threads = 1;
    long npre = 0;
    void clear() {
        for (auto p : pre) delete p;
        for (auto p : post) delete p;
        for (auto p : fr) delete p;
        pre.clear();
        post.clear();
        fr.clear();
        npre = 0;
    }
};

int main() {
    ios::sync_with_stdio(false);
    string line;
    JsonJob job;
    long cur_thread = 0;
    Options split_opt;
    vector<ThreadConfig*> split_cfg;
    while (getline(cin, line)) {
        if (line.empty()) continue;
        vector<string> t = split(line);
        const string& op = t[0];
        if (op == "ED") {
            string a = str(t[1]), b = str(t[2]);
            cout << edit_distance(a.c_str(), a.length(), b.c_str(), b.length()) << "\n";
        } else if (op == "TAC") {
            Options opt;
            int front = atoi(t[1].c_str()), tail = atoi(t[2].c_str());
            opt.qualityCut.enabledFront = atoi(t[3].c_str());
            opt.qualityCut.enabledTail = atoi(t[4].c_str());
            opt.qualityCut.windowSizeFront = atoi(t[5].c_str());
            opt.qualityCut.qualityFront = atoi(t[6].c_str());
            opt.qualityCut.windowSizeTail = atoi(t[7].c_str());
            opt.qualityCut.qualityTail = atoi(t[8].c_str());
            Filter filter(&opt);
            Read r("@n", str(t[9]).c_str(), "+", str(t[10]).c_str());
            int frontTrimmed = 0;
            Read* ret = filter.trimAndCut(&r, front, tail, frontTrimmed);
            if (ret == NULL) cout << "NULL\n";
            else cout << frontTrimmed << " =" << *ret->mSeq << " =" << *ret->mQuality << "\n";
        } else if (op == "PX") {
            Read r("@n", str(t[2]).c_str(), "+", str(t[3]).c_str());
            FilterResult fr(NULL, false);
            PolyX::trimPolyX(&r, &fr, atoi(t[1].c_str()));
            /* per-base counters are private: print them through the reference's own JSON writer */
            cout << "=" << *r.mSeq << " " << fr.getTotalPolyXTrimmedReads() << " "
                 << fr.getTotalPolyXTrimmedBases() << "\n";
        } else if (op == "PF") {
            Options opt;
            opt.qualfilter.enabled = atoi(t[1].c_str());
            opt.qualfilter.qualifiedQual = (char)atoi(t[2].c_str());
            opt.qualfilter.unqualifiedPercentLimit = atoi(t[3].c_str());
            opt.qualfilter.nBaseLimit = atoi(t[4].c_str());
            opt.qualfilter.nBasePercentLimit = atoi(t[5].c_str());
            opt.qualfilter.avgQualReq = atoi(t[6].c_str());
            opt.lengthFilter.enabled = atoi(t[7].c_str());
            opt.lengthFilter.requiredLength = atoi(t[8].c_str());
            opt.lengthFilter.maxLength = atoi(t[9].c_str());
            opt.complexityFilter.enabled = atoi(t[10].c_str());
            /* exactly what src/main.cpp:219 computes from -Y */
            opt.complexityFilter.threshold = (min(100, max(0, atoi(t[11].c_str())))) / 100.0;
            Filter filter(&opt);
            Read r("@n", str(t[12]).c_str(), "+", str(t[13]).c_str());
            cout << filter.passFilter(&r) << "\n";
        } else if (op == "TF" || op == "RS") {
            Read r("@n", str(t[2]).c_str(), "+", str(t[3]).c_str());
            if (op == "TF") r.trimFront(atoi(t[1].c_str()));
            else r.resize(atoi(t[1].c_str()));
            cout << "=" << *r.mSeq << " =" << *r.mQuality << "\n";
        } else if (op == "BG") {
            Read r(str(t[3]).c_str(), str(t[4]).c_str(), str(t[5]).c_str(), str(t[6]).c_str());
            vector<Read*> out = r.breakByGap(atoi(t[1].c_str()), atoi(t[2].c_str()));
            string s;
            for (size_t i = 0; i < out.size(); i++) {
                out[i]->appendToString(&s);
                delete out[i];
            }
            cout << out.size() << " " << s.size() << "\n" << s;
        } else if (op == "LQR") { /* Filter::detectLowQualityRegions(window, quality) */
            Options opt;
            Filter filter(&opt);
            Read r("@n", str(t[3]).c_str(), "+", str(t[4]).c_str());
            vector<pair<int, int>> regions = filter.detectLowQualityRegions(&r, atoi(t[1].c_str()), atoi(t[2].c_str()));
            cout << regions.size();
            for (auto& rg : regions) cout << " " << rg.first << " " << rg.second;
            cout << "\n";
        } else if (op == "BRK") { /* the --break / --mask stage of processSingleEnd (src/seprocessor.cpp:234-262) on
                                     one read: detect + breakByRegions, detect + maskRegionWithN, appendToString */
            Options opt;
            Filter filter(&opt);
            const int be = atoi(t[1].c_str()), bw = atoi(t[2].c_str()), bq = atoi(t[3].c_str());
            const int me = atoi(t[4].c_str()), mw = atoi(t[5].c_str()), mq = atoi(t[6].c_str());
            Read* r1 = new Read(str(t[7]).c_str(), str(t[8]).c_str(), str(t[9]).c_str(), str(t[10]).c_str());
            vector<Read*> outReads;
            outReads.push_back(r1);
            if (be) {
                vector<Read*> tmpReads;
                for (size_t i = 0; i < outReads.size(); i++) {
                    Read* rr = outReads[i];
                    vector<pair<int, int>> regions = filter.detectLowQualityRegions(rr, bw, bq);
                    if (regions.size() > 0) {
                        vector<Read*> brs = rr->breakByRegions(regions);
                        for (size_t j = 0; j < brs.size(); j++) tmpReads.push_back(brs[j]);
                    } else {
                        tmpReads.push_back(rr);
                    }
                }
                outReads = tmpReads;
            }
            if (me) {
                for (size_t i = 0; i < outReads.size(); i++) {
                    Read* rr = outReads[i];
                    vector<pair<int, int>> regions = filter.detectLowQualityRegions(rr, mw, mq);
                    for (size_t j = 0; j < regions.size(); j++)
                        rr->maskRegionWithN(regions[j].first, regions[j].second - regions[j].first + 1);
                }
            }
            string s;
            for (size_t i = 0; i < outReads.size(); i++) outReads[i]->appendToString(&s);
            cout << outReads.size() << " " << s.size() << "\n" << s;
        } else if (op == "TAG") {
            Read r(str(t[2]).c_str(), str(t[3]).c_str(), str(t[4]).c_str(), str(t[5]).c_str());
            string s;
            r.appendToStringWithTag(&s, FAILED_TYPES[atoi(t[1].c_str())]);
            cout << s.size() << "\n" << s;
        } else if (op == "FA") { /* FastaReader::readAll, src/fastareader.cpp:91-101 */
            FastaReader reader(str(t[1]));
            reader.readAll();
            map<string, string> contigs = reader.contigs();
            cout << contigs.size() << "\n";
            for (auto& kv : contigs) cout << "=" << hex_of(kv.first) << " =" << hex_of(kv.second) << "\n";
        } else if (op == "S_BEGIN") {
            for (auto c : split_cfg) delete c;
            split_cfg.clear();
            split_opt = Options();
            split_opt.seqLen = 1000; /* (Options() leaves it unset; main.cpp evaluates it before any Stats exists) */
            split_opt.thread = atoi(t[1].c_str());
            split_opt.split.enabled = true;
            split_opt.split.byFileLines = atoi(t[2].c_str()) != 0;
            split_opt.split.byFileNumber = !split_opt.split.byFileLines;
            split_opt.split.number = atoi(t[3].c_str());
            split_opt.split.size = atol(t[4].c_str());
            split_opt.split.digits = atoi(t[5].c_str());
            split_opt.compression = atoi(t[6].c_str());
            split_opt.out = str(t[7]);
            for (int w = 0; w < split_opt.thread; w++) { /* SingleEndProcessor::initConfig, src/seprocessor.cpp:54-63 */
                ThreadConfig* c = new ThreadConfig(&split_opt, w, false);
                c->initWriterForSplit();
                split_cfg.push_back(c);
            }
            cout << "ok\n";
        } else if (op == "S_PACK") {
            ThreadConfig* c = split_cfg.at(atoi(t[1].c_str()));
            const string text = str(t[3]);
            if (!split_opt.out.empty()) c->getWriter1()->writeString(text); /* src/seprocessor.cpp:297-301 */
            c->markProcessed(atol(t[2].c_str()));                           /* :313-316 */
            cout << (c->canBeStopped() ? 1 : 0) << "\n";
        } else if (op == "S_END") {
            for (auto c : split_cfg) delete c;
            split_cfg.clear();
            cout << "ok\n";
        } else if (op == "J_BEGIN") {
            job.clear();
            job.threads = atoi(t[1].c_str());
            job.opt = Options();
            job.opt.seqLen = atoi(t[2].c_str());
            job.opt.isRNA = atoi(t[3].c_str());
            job.opt.adapter.enabled = atoi(t[4].c_str());
            job.opt.polyXTrim.enabled = atoi(t[5].c_str());
            job.opt.complexityFilter.enabled = atoi(t[6].c_str());
            job.opt.adapter.sequenceStart = str(t[7]);
            job.opt.adapter.sequenceEnd = str(t[8]);
            job.opt.adapter.hasFasta = false;
            for (int i = 0; i < job.threads; i++) { /* what ThreadConfig does, src/threadconfig.cpp:4-17 */
                job.pre.push_back(new Stats(&job.opt));
                job.post.push_back(new Stats(&job.opt));
                job.fr.push_back(new FilterResult(&job.opt));
            }
            cur_thread = 0;
        } else if (op == "J_PRE") {
            /* packs of PACK_SIZE reads go round-robin to the workers, src/seprocessor.cpp:373-378 */
            cur_thread = (job.npre / PACK_SIZE) % job.threads;
            job.npre++;
            Read r("@n", str(t[1]).c_str(), "+", str(t[2]).c_str());
            job.pre[cur_thread]->statRead(&r);
        } else if (op == "J_POST") {
            Read r("@n", str(t[1]).c_str(), "+", str(t[2]).c_str());
            job.post[cur_thread]->statRead(&r);
        } else if (op == "J_FR") {
            job.fr[cur_thread]->addFilterResult(atoi(t[1].c_str()), 1);
        } else if (op == "J_AD") {
            job.fr[cur_thread]->addAdapterTrimmed(str(t[1]));
        } else if (op == "J_ART") {
            job.fr[cur_thread]->addReadTrimmed(atoi(t[1].c_str()));
        } else if (op == "J_PXT") {
            job.fr[cur_thread]->addPolyXTrimmed(atoi(t[1].c_str()), atoi(t[2].c_str()));
        } else if (op == "J_END" || op == "J_ENDH") {
            const bool html = op == "J_ENDH";
            job.opt.jsonFile = str(t[html ? 3 : 1]);
            if (html) {
                job.opt.lengthFilter.enabled = atoi(t[1].c_str());
                job.opt.lengthFilter.maxLength = atoi(t[2].c_str());
                job.opt.htmlFile = str(t[4]);
                string title = str(t[5]);
                for (size_t i = 6; i < t.size(); i++) title += " " + t[i];
                job.opt.reportTitle = title;
            }
            /* what SingleEndProcessor::process does after the join, src/seprocessor.cpp:108-142 */
            Stats* finalPre = Stats::merge(job.pre);
            Stats* finalPost = Stats::merge(job.post);
            if (html) {
                finalPre->calcLengthHistogram();
                finalPost->calcLengthHistogram();
            }
            FilterResult* finalFr = FilterResult::merge(job.fr);
            command = "";
            JsonReporter jr(&job.opt);
            jr.report(finalFr, finalPre, finalPost);
            if (html) {
                HtmlReporter hr(&job.opt);
                hr.report(finalFr, finalPre, finalPost);
            }
            delete finalPre;
            delete finalPost;
            delete finalFr;
            job.clear();
            cout << "OK\n";
        } else {
            cout << "ERR unknown op " << op << "\n";
        }
        cout.flush();
    }
    return 0;
}
